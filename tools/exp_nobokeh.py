import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import ZoicCamera, PRECISION_FAST
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
for use in (True, False):
    c = CONFIGS["C3"]; p = camera_params("C3"); p["useImage"] = use
    cam = ZoicCamera(0)
    if use: cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**p); cam.set_precision(PRECISION_FAST)
    n = ray_count("C3")
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"])
    out = dict(rays=torch.empty((n, 8), device="cuda"))
    for _ in range(3): cam.create_rays(s, out=out)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): cam.create_rays(s, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    cnt = cam.counters()
    print("useImage", use, "ms %.3f Grays/s %.2f" % (dt * 1e3, n / dt / 1e9), "retried frac unknown; vign", cnt["vignettedRays"] / (cnt["succesRays"] + cnt["vignettedRays"]))
