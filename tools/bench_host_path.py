"""PCIe-inclusive rate of the host-buffer API (zoic_create_rays_host): pieces on two streams; pageable vs page-locked
caller buffers (zoic_host_alloc) vs caller memory registered once (zoic_host_register).  Also the per-sample adapter's
latency and the Arnold-layout batch.    python tools/bench_host_path.py [n]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from zoic_amd import PRECISION_FAST, PinnedArray, ZoicCamera, _capi
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, synthetic_samples

def main(n=1 << 24, reps=4):
    c = CONFIGS["C3"]; cam = ZoicCamera(0); cam.set_bokeh_image(hexagon_bokeh()); cam.update(**camera_params("C3")); cam.set_precision(PRECISION_FAST)
    lib = cam._lib
    s = synthetic_samples(n, c["width"], c["height"], c["spp"])
    res = {}
    def timed(sp, rp, tag):
        for _ in range(4):   # the PCIe link / copy engines take a few transfers to reach their steady rate
            lib.zoic_create_rays_host(cam._h, n, sp, None, 0, rp)
        t = time.perf_counter()
        for _ in range(reps):
            cam._check(lib.zoic_create_rays_host(cam._h, n, sp, None, 0, rp))
        dt = (time.perf_counter() - t) / reps
        res[tag] = dict(ms=round(dt * 1e3, 2), grays_s=round(n / dt / 1e9, 3), pcie_gb_s=round(48 * n / dt / 1e9, 1))
        print("%-28s %d rays in %7.1f ms = %.2f Grays/s (%.1f GB/s over PCIe, both directions, incl. kernel)" % (tag, n, dt * 1e3, n / dt / 1e9, 48 * n / dt / 1e9), flush=True)
    rays = np.empty(n, dtype=_capi.RAY_DTYPE)
    timed(s.ctypes.data, rays.ctypes.data, "pageable")
    ps, pr = PinnedArray((n, 4), np.float32), PinnedArray((n,), _capi.RAY_DTYPE)
    ps.array[:] = s
    timed(ps.array.ctypes.data, pr.array.ctypes.data, "pinned (zoic_host_alloc)")
    assert np.array_equal(pr.array.view(np.uint32), rays.view(np.uint32))
    t = time.perf_counter()
    cam._check(lib.zoic_host_register(s.ctypes.data, s.nbytes)); cam._check(lib.zoic_host_register(rays.ctypes.data, rays.nbytes))
    treg = time.perf_counter() - t
    print("zoic_host_register of %.0f MB: %.1f ms (%.1f GB/s)" % ((s.nbytes + rays.nbytes) / 1e6, treg * 1e3, (s.nbytes + rays.nbytes) / treg / 1e9))
    res["register_ms"] = round(treg * 1e3, 2)
    timed(s.ctypes.data, rays.ctypes.data, "registered caller memory")
    lib.zoic_host_unregister(s.ctypes.data); lib.zoic_host_unregister(rays.ctypes.data)
    # Arnold layout batch
    m = 1 << 22
    inp = np.zeros((m, 7), np.float32); inp[:, [0, 1, 4, 5]] = s[:m]
    cam.create_rays_arnold(inp[:1000])
    t = time.perf_counter(); cam.create_rays_arnold(inp); dt = time.perf_counter() - t
    res["arnold_batch"] = dict(n=m, ms=round(dt * 1e3, 2), mrays_s=round(m / dt / 1e6, 1))
    print("Arnold-layout batch: %d samples in %.1f ms = %.1f Mrays/s (28 B in, 84 B out per sample on the host side)" % (m, dt * 1e3, m / dt / 1e6))
    # per-sample adapter latency
    i = _capi.CameraInput(0.1, 0.05, 0, 0, 0.3, 0.6, 0); o = _capi.CameraOutput()
    for _ in range(200): lib.zoic_camera_create_ray(cam._h, C.byref(i), C.byref(o), 1)
    k = 5000; t = time.perf_counter()
    for _ in range(k): lib.zoic_camera_create_ray(cam._h, C.byref(i), C.byref(o), 1)
    dt = (time.perf_counter() - t) / k
    res["per_sample_us"] = round(dt * 1e6, 2)
    print("camera_create_ray per-sample adapter: %.1f us per call (launch + stream sync, zero-copy sample/record)" % (dt * 1e6))
    print("HOSTPATH_JSON " + json.dumps(res))
    return res

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24)
