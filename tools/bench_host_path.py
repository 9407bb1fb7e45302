"""PCIe-inclusive rate of the host-buffer API (zoic_create_rays_host): H2D + kernel + D2H, synchronous."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zoic_amd import ZoicCamera, PRECISION_FAST
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, synthetic_samples
c = CONFIGS["C3"]; cam = ZoicCamera(0); cam.set_bokeh_image(hexagon_bokeh()); cam.update(**camera_params("C3")); cam.set_precision(PRECISION_FAST)
n = 1 << 24
s = synthetic_samples(n, c["width"], c["height"], c["spp"])
cam.create_rays(s[:1024])
import ctypes as C
from zoic_amd import _capi
rays = np.empty(n, dtype=_capi.RAY_DTYPE)
t = time.perf_counter()
for _ in range(3):
    cam._check(cam._lib.zoic_create_rays_host(cam._h, n, s.ctypes.data, None, 0, rays.ctypes.data))
dt = (time.perf_counter() - t) / 3
print("host-buffer path: %d rays in %.1f ms = %.2f Grays/s (%.1f GB/s over PCIe incl. kernel)" % (n, dt * 1e3, n / dt / 1e9, 48 * n / dt / 1e9))
