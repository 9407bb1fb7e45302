"""tools/exp_thin_vignet.py: thin-lens model with empirical optical vignetting (zoic.cpp:1804-1819 retry loop): Grays/s vs distance."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import ZoicCamera, PRECISION_FAST
from zoic_amd.workloads import CONFIGS, camera_params
c = CONFIGS["C3"]
n = c["width"] * c["height"] * c["spp"]
for ov in (0.0, 1.0, 2.0, 5.0, 10.0, 30.0):
    p = dict(camera_params("C1"), opticalVignettingDistance=ov)
    cam = ZoicCamera(0); cam.update(**p); cam.set_precision(PRECISION_FAST)
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"])
    out = dict(rays=torch.empty((n, 8), device="cuda"))
    for _ in range(2): cam.create_rays(s, out=out)
    cam.reset_counters()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): cam.create_rays(s, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    cnt = cam.counters()
    retried = float(((out["rays"][:, 7].view(torch.int32) & 1) != 0).float().mean().item())
    print("opticalVignettingDistance %5.1f: %7.3f ms  %6.1f Grays/s   retried %.3f  zero-weight %.4f" % (
        ov, dt * 1e3, n / dt / 1e9, retried, cnt["vignettedRays"] / (cnt["succesRays"] + cnt["vignettedRays"])))
