"""tools/exp_thin_image.py: thin-lens model WITH a bokeh image (no bench config covers it): Grays/s at a 132.7 M-ray frame."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import ZoicCamera, PRECISION_FAST
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh
c = CONFIGS["C3"]; p = dict(camera_params("C1"), useImage=True, bokehPath="procedural:hexagon256")
for ov in (0.0, 30.0):
    p["opticalVignettingDistance"] = ov
    cam = ZoicCamera(0); cam.set_bokeh_image(hexagon_bokeh()); cam.update(**p); cam.set_precision(PRECISION_FAST)
    n = c["width"] * c["height"] * c["spp"]
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"])
    out = dict(rays=torch.empty((n, 8), device="cuda"))
    for _ in range(3): cam.create_rays(s, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): cam.create_rays(s, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("thin lens + bokeh image, opticalVignettingDistance %g: %.3f ms  %.1f Grays/s  %.2f TB/s algorithmic" % (ov, dt * 1e3, n / dt / 1e9, n * 48 / dt / 1e12))
