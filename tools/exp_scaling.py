"""tools/exp_scaling.py [C2] [mode]: launch time against batch size at a fixed spatial distribution (spp varied): what is fixed per launch?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zoic_amd import ZoicCamera, PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh
cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
mode = sys.argv[2] if len(sys.argv) > 2 else "unchecked"
c = CONFIGS[cfg]
cam = ZoicCamera(0)
if c["bokeh"]:
    cam.set_bokeh_image(hexagon_bokeh())
cam.update(**camera_params(cfg))
cam.set_precision({"fast": PRECISION_FAST, "unchecked": PRECISION_FAST_UNCHECKED, "strict": PRECISION_STRICT}[mode])
for spp in (1, 2, 4, 8, 16, 32, 64):
    n = c["width"] * c["height"] * spp
    if n > 600_000_000:
        break
    s = cam.generate_samples(n, c["width"], c["height"], spp, seed=1)
    out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device="cuda"))
    for _ in range(3):
        cam.create_rays(s, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cam.create_rays(s, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s %s spp %2d n=%10d %8.4f ms %6.2f Grays/s" % (cfg, mode, spp, n, ms, n / ms / 1e6), flush=True)
    del s, out
