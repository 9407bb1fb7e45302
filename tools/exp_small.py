"""tools/exp_small.py [C2] [mode]: launch time of small batches (samples spread over the whole frame)"""
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
full = cam.generate_samples(c["width"] * c["height"], c["width"], c["height"], 1, seed=1)
for n in (4096, 65536, 262144, 524288, 1048576, 2073600):
    stride = full.shape[0] // n
    s = full[::stride][:n].contiguous()
    out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device="cuda"))
    for _ in range(3):
        cam.create_rays(s, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        cam.create_rays(s, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s %s n=%10d %8.4f ms %6.2f Grays/s" % (cfg, mode, n, ms, n / ms / 1e6), flush=True)
