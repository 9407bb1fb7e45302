"""The C-ABI under the reference's calling contract: camera_create_ray is called concurrently from every
render thread with a `tid` (zoic.cpp:1752); node_update is not.  Everything here goes through ctypes -> libzoic_amd.so;
ctypes releases the GIL around the foreign call, so the Python threads below really do run the entry points in parallel.
"""
import threading

import numpy as np
import pytest

from zoic_amd import PRECISION_STRICT, PinnedArray, ZoicCamera, ZoicError
from zoic_amd import _capi
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_rng_states, synthetic_samples

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _c2_camera():
    cam = ZoicCamera(0)
    cam.update(**camera_params("C2"))
    cam.set_precision(PRECISION_STRICT)
    return cam


def _slab(cfg, n, where):
    c = CONFIGS[cfg]
    base = int(c["width"] * int(c["height"] * where)) * c["spp"]
    return synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base), base


def _out_tuple(o):
    return (o.origin.x, o.origin.y, o.origin.z, o.dir.x, o.dir.y, o.dir.z, o.weight[0], o.dOdy.x, o.dDdy.z)


def test_one_render_thread_replays_the_reference_process(gpu, oracle_lib):
    """tid 0's retry stream is the reference's process-global xor128 state (advanced by node_update's LUT build, then by
    every retry in sample order): per-sample calls from ONE thread must equal the oracle's sequential run, bit for bit,
    including across a second node_update (zoic.cpp:647-652, 1411-1412, 1930)."""
    cam = _c2_camera()
    oc = oracle_lib.OracleCamera()
    oc.update(**camera_params("C2"))
    n = 1920
    s, _ = _slab("C2", 8 * n, 0.2)
    s = s[::8]                            # one sample per pixel of a whole row: first-try, retried and zero-weight rays
    ref = oc.create_rays(s)               # rng_states=None: the sequential global stream
    assert (ref["flags"] & 1).mean() > 0.1 and ((ref["tries"] > 0) & (ref["weight"] != 0)).sum() > 20
    got = np.array([_out_tuple(cam.create_ray(*[float(v) for v in row], tid=0)) for row in s], np.float32)
    assert np.array_equal(bits(got[:, 0:3].T.copy()), bits(ref["origin"]))
    assert np.array_equal(bits(got[:, 3:6].T.copy()), bits(ref["dir"]))
    assert np.array_equal(got[:, 6], ref["weight"])
    # a lens change re-runs the LUT build from wherever the stream stands now -- in both implementations
    p2 = dict(camera_params("C2"), fStop=4.0)
    cam.update(**p2)
    oc.update(**p2)
    assert np.array_equal(cam.info()["lutBoxes"], oc.lut()[1])
    ref2 = oc.create_rays(s[:300])
    got2 = np.array([_out_tuple(cam.create_ray(*[float(v) for v in row], tid=0)) for row in s[:300]], np.float32)
    assert np.array_equal(bits(got2[:, 3:6].T.copy()), bits(ref2["dir"]))


def test_per_sample_fuzz_replays_the_reference_process(gpu, oracle_lib):
    """The per-sample call through the resident kernel (csrc/mailbox.hip: its own one-ray code path, not the batch kernels)
    behind machine-made cameras: every shipped prescription with a stop, both lens models, LUT on / off, bokeh images that
    are not square, some samples nobody should send.  240 calls from render thread 0 must equal the oracle's sequential
    run (the process-global xor128 stream, advanced by the LUT build and by every retry in sample order), bit for bit."""
    import os
    from hypothesis import given, settings, HealthCheck, strategies as st
    from zoic_amd import RAYTRACED, THINLENS, lens_path
    lenses = ["double_gauss_f2.0.dat", "tessar_f2.8.dat", "fisheye_muller_f4.0.dat", "petzval_f1.25.dat", "triplet_f2.5.dat", "mori_f2.8.dat"]
    special = np.array([0.0, -0.0, 0.5, 1.0, -1.0, 0.99999994, 1e-40, 1e30, np.inf, -np.inf, np.nan, 2.0, -3.0], np.float32)

    @settings(max_examples=int(os.environ.get("ZOIC_FUZZ_EXAMPLES_SAMPLE", os.environ.get("ZOIC_FUZZ_EXAMPLES", "40"))), deadline=None,
              suppress_health_check=list(HealthCheck), derandomize=True)
    @given(st.sampled_from(lenses), st.sampled_from([RAYTRACED, RAYTRACED, RAYTRACED, THINLENS]), st.booleans(), st.booleans(),
           st.floats(2.0, 12.0, width=32), st.floats(1.25, 11.0, width=32), st.floats(1.0, 7.5, width=32), st.integers(0, 2 ** 16), st.floats(0.05, 0.95))
    def run(lens, model, lut, image, focal, fstop, sensor_w, seed, where):
        rs = np.random.RandomState(seed)
        p = dict(lensModel=model, lensDataPath=lens_path(lens), focalLength=focal, fStop=fstop, focalDistance=100.0, sensorWidth=sensor_w,
                 sensorHeight=sensor_w / 1.5, kolbSamplingLUT=lut, useImage=image, bokehPath="mem:persample%d" % seed)
        cam, oc = ZoicCamera(0), oracle_lib.OracleCamera()
        if image:
            h, w = int(rs.randint(2, 24)), int(rs.randint(2, 24))
            img = np.repeat(rs.rand(h, w).astype(np.float32)[:, :, None], 3, axis=2)
            cam.set_bokeh_image(img); oc.set_bokeh_image(img)
        try:
            oc.update(**p)
        except oracle_lib.OracleError:
            return
        cam.update(**p)
        n = 240
        s, _ = _slab("C2", 8 * n, where)
        s = s[::8].copy()
        hostile = rs.rand(n, 4) < 0.02
        s[hostile] = special[rs.randint(len(special), size=int(hostile.sum()))]
        ref = oc.create_rays(s)               # rng_states=None: the sequential global stream
        got = np.array([_out_tuple(cam.create_ray(*[float(v) for v in row], tid=0)) for row in s], np.float32)
        for a, b in ((got[:, 0:3].T.copy(), ref["origin"]), (got[:, 3:6].T.copy(), ref["dir"]), (got[:, 6].copy(), ref["weight"])):
            same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
            assert same.all(), (p, np.nonzero(~same)[-1][:4])
    run()


def test_update_sequence_fuzz(gpu, oracle_lib):
    """node_update is stateful (zoic.cpp:1575-1720): the lens is rebuilt -- and the process-global xor128 stream advanced
    by the LUT build's 6.4 M draws -- only when a lens parameter changed (1615, 1708-1710), the bokeh tables only when the
    image switch or path did, and everything else is re-derived every time.  Random sequences of updates on ONE camera
    (f-stop, focal length, focus, sensor, LUT / DOF / image / lens-model switches, another prescription, another image,
    an update that changes nothing) against ONE oracle camera: after every step the exit-pupil LUT, a batch of rays (per-ray
    streams) and a run of per-sample calls on render thread 0 (the global stream: it only stays in step if both sides
    rebuilt -- or skipped -- the same things) must be bit-identical."""
    import os
    from hypothesis import given, settings, HealthCheck, strategies as st
    from zoic_amd import RAYTRACED, THINLENS, lens_path
    lenses = ["double_gauss_f2.0.dat", "tessar_f2.8.dat", "fisheye_muller_f4.0.dat", "petzval_f1.25.dat", "triplet_f2.5.dat", "mori_f2.8.dat"]
    ops = ["fstop", "focal", "focus", "sensor", "lut", "model", "image", "newimage", "lens", "same", "exposure", "dof", "ov"]

    @settings(max_examples=int(os.environ.get("ZOIC_FUZZ_EXAMPLES_UPDATE", os.environ.get("ZOIC_FUZZ_EXAMPLES", "25"))), deadline=None,
              suppress_health_check=list(HealthCheck), derandomize=True)
    @given(st.integers(0, 2 ** 16), st.lists(st.sampled_from(ops), min_size=3, max_size=7))
    def run(seed, steps):
        rs = np.random.RandomState(seed)
        cam, oc = ZoicCamera(0), oracle_lib.OracleCamera()
        images = [0]

        def new_image():
            images[0] += 1
            h, w = int(rs.randint(2, 48)), int(rs.randint(2, 48))
            img = np.repeat(rs.rand(h, w).astype(np.float32)[:, :, None], 3, axis=2)
            cam.set_bokeh_image(img); oc.set_bokeh_image(img)
            return "mem:seq%d_%d" % (seed, images[0])

        p = dict(lensModel=RAYTRACED, lensDataPath=lens_path(lenses[rs.randint(len(lenses))]), focalLength=5.0, fStop=2.8, focalDistance=100.0,
                 sensorWidth=3.6, sensorHeight=2.4, kolbSamplingLUT=True, useImage=False, bokehPath="", useDof=True, exposureControl=0.0,
                 opticalVignettingDistance=0.0)
        for step, op in enumerate(["first"] + list(steps)):
            if op == "fstop": p["fStop"] = float(np.float32(rs.uniform(1.4, 11.0)))
            elif op == "focal": p["focalLength"] = float(np.float32(rs.uniform(3.0, 10.0)))
            elif op == "focus": p["focalDistance"] = float(np.float32(rs.uniform(30.0, 500.0)))
            elif op == "sensor": p["sensorWidth"] = float(np.float32(rs.uniform(1.5, 5.0)))
            elif op == "lut": p["kolbSamplingLUT"] = not p["kolbSamplingLUT"]
            elif op == "model": p["lensModel"] = THINLENS if p["lensModel"] == RAYTRACED else RAYTRACED
            elif op == "image":
                p["useImage"] = not p["useImage"]
                if p["useImage"] and not p["bokehPath"]:
                    p["bokehPath"] = new_image()
            elif op == "newimage":
                p["bokehPath"] = new_image(); p["useImage"] = True
            elif op == "lens": p["lensDataPath"] = lens_path(lenses[rs.randint(len(lenses))])
            elif op == "exposure": p["exposureControl"] = float(np.float32(rs.uniform(-2.0, 2.0)))
            elif op == "dof": p["useDof"] = not p["useDof"]
            elif op == "ov": p["opticalVignettingDistance"] = float(np.float32(rs.uniform(0.0, 5.0)))
            perr = oerr = None
            try:
                cam.update(**p)
            except ZoicError as e:
                perr = getattr(e, "status_name", type(e).__name__).replace("ZOIC_ERR_", "")
            try:
                oc.update(**p)
            except oracle_lib.OracleError as e:
                oerr = oracle_lib.ERR_NAMES[e.code]
            assert perr == oerr, (step, op, p, perr, oerr)
            if perr is not None:
                return
            if p["lensModel"] == RAYTRACED and p["kolbSamplingLUT"]:
                assert np.array_equal(bits(cam.info()["lutBoxes"]), bits(oc.lut()[1])), (step, op, p)
            n = 2048
            s, base = _slab("C2", n, float(rs.uniform(0.05, 0.95)))
            ref = oc.create_rays(s, rng_states=ray_rng_states(n, seed=1, ray_index_base=base), threads=4)
            got = cam.create_rays(s, ray_index_base=base)
            assert np.array_equal(got["flags"], ref["flags"]), (step, op, p)
            g, r = got["planes"], ref["planes"]
            assert ((bits(g) == bits(r)) | (np.isnan(g) & np.isnan(r))).all(), (step, op, p)
            t = s[::64][:32]
            seq = oc.create_rays(t)            # the sequential global stream
            one = np.array([_out_tuple(cam.create_ray(*[float(v) for v in row], tid=0)) for row in t], np.float32)
            for a, b in ((one[:, 0:3].T.copy(), seq["origin"]), (one[:, 3:6].T.copy(), seq["dir"]), (one[:, 6].copy(), seq["weight"])):
                assert ((bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))).all(), (step, op, p)
    run()


def test_mailbox_survives_idle_lifetime_mode_changes_and_shared_slots(gpu, oracle_lib):
    """The resident per-sample kernel (csrc/mailbox.hip) retires after 1 ms without a call and after 50 ms in any case, is
    stopped by set_precision / update / the counter getters, and serves tids 64 apart from ONE slot.  Whatever it does, the
    rays of a tid are those of that tid's retry stream: tid 0 = the oracle's sequential run (a fresh camera per leg), the
    other tids = a per-tid replay on a second camera that is never disturbed."""
    import time
    from zoic_amd import PRECISION_FAST
    oc = oracle_lib.OracleCamera()
    oc.update(**camera_params("C2"))
    s, _ = _slab("C2", 8 * 700, 0.2)
    s = s[::8]
    ref = oc.create_rays(s)                                    # tid 0: the sequential global stream
    cam = _c2_camera()
    got = []
    for k, row in enumerate(s):
        if k in (50, 120, 121, 400):
            time.sleep(0.02)                                   # > 1 ms idle: the kernel has retired, the call restarts it
        if k == 200:
            cam.set_precision(PRECISION_FAST)                  # stops the kernel; STRICT again before the next call
            cam.set_precision(PRECISION_STRICT)
        if k == 300:
            assert cam.counters()["succesRays"] + cam.counters()["vignettedRays"] == 300   # stops it, its counts arrive
        got.append(_out_tuple(cam.create_ray(*[float(v) for v in row], tid=0)))
    got = np.array(got, np.float32)
    assert np.array_equal(bits(got[:, 3:6].T.copy()), bits(ref["dir"])) and np.array_equal(got[:, 6], ref["weight"])
    # > 50 ms of uninterrupted calls (the lifetime cap restarts the kernel in mid-flight), three tids on one slot
    plain = _c2_camera()
    tids = (3, 67, 131, 3, 131, 67)
    rows = [[float(v) for v in s[(7 * i) % len(s)]] for i in range(6000)]
    a = [_out_tuple(cam.create_ray(*rows[i], tid=tids[i % 6])) for i in range(6000)]
    want = {}
    for t in (3, 67, 131):                                     # the same calls, one tid after the other, on an undisturbed camera
        for i in range(6000):
            if tids[i % 6] == t:
                want[i] = _out_tuple(plain.create_ray(*rows[i], tid=t))
    assert all(a[i] == want[i] for i in range(6000))
    total = cam.counters()
    assert total["succesRays"] + total["vignettedRays"] == 700 + 6000
    cam.close(); plain.close()


def test_two_per_sample_calls_that_retry_draw_different_numbers(gpu):
    """Round 1 keyed every per-sample call to ray index 0: all retried samples of a frame drew the same (u, v) sequence.
    Now the tid's stream carries over, so the same sample submitted twice retries with different draws."""
    cam = _c2_camera()
    s, _ = _slab("C2", 16000, 0.2)
    first = cam.create_rays(s)
    pick = (first["tries"] > 0) & (first["weight"] != 0)               # samples that retry and then succeed
    assert pick.sum() > 100
    k = int(np.argmax(pick))
    row = [float(v) for v in s[k]]
    a = _out_tuple(cam.create_ray(*row, tid=5))
    b = _out_tuple(cam.create_ray(*row, tid=5))
    c = _out_tuple(cam.create_ray(*row, tid=6))
    assert a != b and a != c
    assert a[7:] != (0.0, 0.0)             # retried => dOdy/dDdy written (zoic.cpp:1974-1977)
    # same tid history on a fresh camera => same rays (deterministic per thread)
    cam2 = _c2_camera()
    assert _out_tuple(cam2.create_ray(*row, tid=5)) == a
    assert _out_tuple(cam2.create_ray(*row, tid=5)) == b


def test_sixteen_threads_of_mixed_calls_equal_the_serial_result(gpu):
    """16 host threads x 1000 mixed camera_create_ray / batched Arnold-layout / host-buffer calls on ONE camera.
    Each thread's results must equal what the same call sequence produces when the threads run one after another."""
    n_threads, n_calls = 16, 1000
    s_all, base = _slab("C2", 1 << 16, 0.04)

    def work(cam, t, sink):
        rng = np.random.default_rng(1000 + t)
        out = []
        for i in range(n_calls):
            kind = rng.integers(0, 10)
            if kind < 7:                                    # the per-sample callback with this thread's tid
                row = s_all[rng.integers(0, len(s_all))]
                out.append(np.array(_out_tuple(cam.create_ray(*[float(v) for v in row], tid=t)), np.float32))
            elif kind < 9:                                  # a bucket of samples in AtCameraInput layout
                m = int(rng.integers(1, 3000))
                lo = int(rng.integers(0, len(s_all) - m))
                inp = np.zeros((m, 7), np.float32)
                inp[:, [0, 1, 4, 5]] = s_all[lo:lo + m]
                out.append(cam.create_rays_arnold(inp, ray_index_base=base + lo).ravel())
            else:                                           # host-buffer batch
                m = int(rng.integers(1, 20000))
                lo = int(rng.integers(0, len(s_all) - m))
                r = cam.create_rays(s_all[lo:lo + m], ray_index_base=base + lo)
                out.append(np.concatenate([r["planes"].ravel(), r["flags"].astype(np.float32)]))
        sink[t] = np.concatenate(out)

    serial, parallel = {}, {}
    cam = _c2_camera()
    for t in range(n_threads):
        work(cam, t, serial)
    done_serial = cam.counters()
    cam.close()
    cam = _c2_camera()
    errors = []

    def guarded(t):
        try:
            work(cam, t, parallel)
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=guarded, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(n_threads):
        assert np.array_equal(bits(serial[t]), bits(parallel[t])), "thread %d differs from its serial run" % t
    assert cam.counters() == done_serial                   # the shared counters saw every ray exactly once
    cam.close()


def test_more_launches_in_flight_than_launch_slots(gpu):
    """200 asynchronous launches on 8 streams with nothing waited for in between: more than the 64 launch slots, so slots
    are reused behind their completion events.  Every batch must come out as when it runs alone."""
    import torch
    cam = ZoicCamera(0)
    cam.update(**camera_params("C2"))
    c = CONFIGS["C2"]
    sizes = [50_000 + 7_919 * (i % 13) for i in range(200)]
    bases = [c["width"] * 40 * c["spp"] + 100_003 * i for i in range(200)]
    samples = [cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=3, ray_index_base=b) for n, b in zip(sizes, bases)]
    torch.cuda.synchronize()
    alone = []
    for s, b in zip(samples, bases):
        alone.append(cam.create_rays(s, ray_index_base=b)["rays"].clone())
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(8)]
    outs = [dict(rays=torch.empty((n, 8), dtype=torch.float32, device="cuda")) for n in sizes]
    for i, (s, b) in enumerate(zip(samples, bases)):
        cam.create_rays(s, ray_index_base=b, out=outs[i], stream=streams[i % 8].cuda_stream)
    torch.cuda.synchronize()
    for i in range(200):
        assert torch.equal(outs[i]["rays"].view(torch.int32), alone[i].view(torch.int32)), "batch %d differs" % i


def test_reverse_ray_is_false_like_the_reference(gpu):
    cam = _c2_camera()
    assert cam.reverse_ray((1.0, 2.0, 3.0), fov=0.5) is False      # zoic.cpp:1992-1995


@pytest.mark.parametrize("n", [1, 777, 300_000, 5_000_000])
def test_host_path_pieces_pinned_and_pageable_equal_the_device_path(gpu, n):
    """zoic_create_rays_host cuts a call into pieces on two streams; pageable and page-locked caller buffers, with and
    without caller-supplied stream states, must give the device path's rays."""
    import torch
    cam = ZoicCamera(0)
    cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params("C3"))
    c = CONFIGS["C3"]
    base = c["width"] * 900 * c["spp"]
    s = synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    dev = cam.create_rays(torch.from_numpy(s).cuda(), ray_index_base=base)["rays"].cpu().numpy()
    pageable = cam.create_rays(s, ray_index_base=base)
    assert np.array_equal(pageable["rays"].view(np.uint32).reshape(n, 8), dev.view(np.uint32))
    pin_s, pin_r = PinnedArray((n, 4), np.float32), PinnedArray((n,), _capi.RAY_DTYPE)
    pin_s.array[:] = s
    pinned = cam.create_rays(pin_s.array, ray_index_base=base, out=pin_r.array)
    assert np.array_equal(pinned["rays"].view(np.uint32).reshape(n, 8), dev.view(np.uint32))
    st = ray_rng_states(n, seed=9, ray_index_base=base)               # caller-supplied streams ride the same pieces
    with_states = cam.create_rays(s, rng_states=st, ray_index_base=base)
    dev_states = cam.create_rays(torch.from_numpy(s).cuda(), rng_states=torch.from_numpy(st.view(np.int32)).cuda(),
                                 ray_index_base=base)["rays"].cpu().numpy()
    assert np.array_equal(with_states["rays"].view(np.uint32).reshape(n, 8), dev_states.view(np.uint32))
    pin_s.free(); pin_r.free()


def test_failed_bokeh_load_is_not_forgotten(gpu):
    """after a failed bokeh load a second update with the same parameters must not report success with the
    image silently off; new pixels under an unchanged bokehPath must rebuild the CDFs."""
    cam = ZoicCamera(0)
    p = dict(camera_params("C3"))
    with pytest.raises(ZoicError) as e:
        cam.update(**p)                      # useImage with no pixels and a path that is not a .pfm file
    assert e.value.status_name == "ZOIC_ERR_BOKEH_IMAGE"
    with pytest.raises(ZoicError):
        cam.update(**p)                      # round 1: returned ZOIC_OK here and sampled the disk instead
    s, base = _slab("C3", 1000, 0.5)
    with pytest.raises(ZoicError) as e:
        cam.create_rays(s)
    assert e.value.status_name == "ZOIC_ERR_NOT_UPDATED"
    img = hexagon_bokeh()
    cam.set_bokeh_image(img)
    cam.update(**p)
    a = cam.create_rays(s, ray_index_base=base)
    cam.set_bokeh_image(np.ascontiguousarray(img[::-1, :, :] ** 2))   # same path, new pixels
    cam.update(**p)
    b = cam.create_rays(s, ray_index_base=base)
    assert not np.array_equal(bits(a["planes"]), bits(b["planes"]))


def test_non_monotone_cdf_keeps_the_reference_search(gpu, oracle_lib):
    """negative luminance (HDR images) makes the CDFs non-monotone; counting entries <= u is then not
    std::upper_bound.  Such tables must fall back to the reference's own binary search: strict stays bit-exact."""
    rng = np.random.default_rng(5)
    img = rng.random((40, 56, 3)).astype(np.float32)
    img[rng.random((40, 56)) < 0.2] *= -0.5
    p = dict(camera_params("C1"), useImage=True, bokehPath="mem:neg")
    cam, oc = ZoicCamera(0), oracle_lib.OracleCamera()
    cam.set_bokeh_image(img); oc.set_bokeh_image(img)
    cam.update(**p); oc.update(**p)
    t = cam.bokeh_tables()
    assert (np.diff(t["cdfRow"]) < 0).any() or (np.diff(t["cdfColumn"].reshape(40, 56), axis=1) < 0).any()
    n = 1 << 15
    s, base = _slab("C1", n, 0.5)
    got = cam.create_rays(s, ray_index_base=base)
    ref = oc.create_rays(s, rng_states=ray_rng_states(n, 1, base))
    assert np.array_equal(got["flags"], ref["flags"])
    assert np.array_equal(bits(got["planes"]), bits(ref["planes"]))


def test_current_device_is_left_alone(gpu):
    """Entry points run on the camera's device and restore the caller's current device."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    before = ctypes.c_int(-1)
    hip.hipGetDevice(ctypes.byref(before))
    cam = _c2_camera()
    s, _ = _slab("C2", 100, 0.5)
    cam.create_rays(s)
    cam.counters()
    after = ctypes.c_int(-1)
    hip.hipGetDevice(ctypes.byref(after))
    assert before.value == after.value
    import torch
    if torch.cuda.device_count() > 1:
        with pytest.raises(ValueError):
            cam.create_rays(torch.zeros((4, 4), device="cuda:1"))


@pytest.mark.parametrize("cfg", ["C2", "C3", "C4"])
def test_per_sample_call_equals_the_batch_ray_of_the_same_stream_in_both_modes(gpu, cfg):
    """One kernel family, one arithmetic per mode: the FIRST call of a fresh tid t draws its retries from the stream keyed by
    (0xA7100000 | t) << 32 (capi.cpp tid_state), which is the batch path's stream of ray index (0xA7100000 | t) << 32 -- so that
    call and a one-ray batch launch at that ray_index_base must produce the same ray, bit for bit, in STRICT *and* in FAST
    (round 4: the FAST arithmetic is written with explicit FMAs, so the branchy trace of the per-sample kernel, the unrolled
    trace of the batch kernels and the listed kernel's rule for rays too close to call all round alike)."""
    from zoic_amd import PRECISION_FAST
    c = CONFIGS[cfg]
    s, _ = _slab(cfg, 4096, 0.37)
    for mode in (PRECISION_STRICT, PRECISION_FAST):
        cam = ZoicCamera(0)
        if c["bokeh"]:
            cam.set_bokeh_image(hexagon_bokeh())
        cam.update(**camera_params(cfg))
        cam.set_precision(mode)
        first = cam.create_rays(s)
        retried = np.nonzero(first["tries"] > 0)[0][:40]              # rays that draw from their stream ...
        plain = np.nonzero(first["tries"] == 0)[0][:20]               # ... and some that do not
        for t, k in enumerate(list(retried) + list(plain), start=1):
            row = [float(v) for v in s[k]]
            one = cam.create_ray(*row, tid=t)                          # the tid's first call: its stream is at its seed
            got = cam.create_rays(s[k:k + 1], ray_index_base=(0xA7100000 | t) << 32)
            want = np.array([got["planes"][i, 0] for i in range(6)], np.float32)
            have = np.array([one.origin.x, one.origin.y, one.origin.z, one.dir.x, one.dir.y, one.dir.z], np.float32)
            assert np.array_equal(bits(have), bits(want)), (cfg, mode, t, have, want)
            assert np.float32(one.weight[0]) == got["planes"][6, 0]
        cam.close()
