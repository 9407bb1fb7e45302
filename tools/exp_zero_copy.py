"""Experiment: host-buffer path variants on page-locked caller memory (C3 camera, fast mode, 16.8 M samples).
(a) zoic_create_rays_host: pieces on two streams (H2D copy, kernels, D2H copy);
(b) zero-copy: ONE launch whose kernels read the samples from, and write the records to, the mapped host memory directly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
from zoic_amd import PRECISION_FAST, PinnedArray, ZoicCamera, _capi
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, synthetic_samples
c = CONFIGS["C3"]; cam = ZoicCamera(0); cam.set_bokeh_image(hexagon_bokeh()); cam.update(**camera_params("C3")); cam.set_precision(PRECISION_FAST)
n = 1 << 24
s = synthetic_samples(n, c["width"], c["height"], c["spp"])
ps, pr = PinnedArray((n, 4), np.float32), PinnedArray((n,), _capi.RAY_DTYPE)
ps.array[:] = s
lib = cam._lib
def t(fn, reps=4):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for piece in (0, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
    os.environ["ZOIC_HOST_PIECE"] = str(piece) if piece else ""
a = t(lambda: cam._check(lib.zoic_create_rays_host(cam._h, n, ps.array.ctypes.data, None, 0, pr.array.ctypes.data)))
print("pieces on two streams: %.2f ms = %.2f Grays/s (%.1f GB/s both directions)" % (a * 1e3, n / a / 1e9, 48 * n / a / 1e9))
ref = pr.array.copy()
pr.array[:] = 0
def zc():
    cam._check(lib.zoic_create_rays_device(cam._h, n, ps.array.ctypes.data, None, 0, pr.array.ctypes.data, None))
    torch.cuda.synchronize()
b = t(zc)
print("zero-copy single launch: %.2f ms = %.2f Grays/s (%.1f GB/s both directions)" % (b * 1e3, n / b / 1e9, 48 * n / b / 1e9))
print("identical:", np.array_equal(ref.view(np.uint32), pr.array.view(np.uint32)))
# H2D copy + kernel writing records straight to host memory
import torch
d_s = torch.empty((n, 4), dtype=torch.float32, device="cuda")
def h2d_then_zero_copy_out():
    d_s.copy_(torch.from_numpy(ps.array), non_blocking=True)
    cam._check(lib.zoic_create_rays_device(cam._h, n, d_s.data_ptr(), None, 0, pr.array.ctypes.data, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
e = t(h2d_then_zero_copy_out)
print("H2D copy + records written straight to host: %.2f ms = %.2f Grays/s" % (e * 1e3, n / e / 1e9))
