"""C3 with bokeh images of different sizes (the column cell-record table grows with size^2): is the sampler's gather what holds C3 back?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import PRECISION_FAST_UNCHECKED, PRECISION_FAST, ZoicCamera
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
c = CONFIGS["C3"]; n = ray_count("C3")
for size in (0, 16, 64, 256, 1024, 2048):
    cam = ZoicCamera(0)
    p = dict(camera_params("C3"))
    if size:
        cam.set_bokeh_image(hexagon_bokeh(size)); p["bokehPath"] = "mem:%d" % size
    else:
        p["useImage"] = False
    cam.update(**p)
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1)
    out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device="cuda"))
    for mode, name in ((PRECISION_FAST_UNCHECKED, "unchecked"),):
        cam.set_precision(mode)
        for _ in range(2): cam.create_rays(s, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): cam.create_rays(s, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        cnt = cam.counters()
        print("bokeh %4d^2 %s: %.3f ms %.2f Grays/s  zero-weight %.4f" % (size, name, ms, n / ms / 1e6, cnt["vignettedRays"] / max(1, cnt["vignettedRays"] + cnt["succesRays"])), flush=True)
    cam.close(); del s, out; torch.cuda.empty_cache()
