"""node_update timing (cold path): lens precompute + exit-pupil LUT (GPU probes vs host) and the bokeh CDF build by image size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zoic_amd import ZoicCamera, THINLENS
from zoic_amd.workloads import camera_params, hexagon_bokeh

def t(f, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3

for cfg in ("C2", "C4", "C5"):
    p = camera_params(cfg)
    res = []
    for mode in (None, "2", "1"):
        os.environ.pop("ZOIC_LUT_HOST", None)
        if mode:
            os.environ["ZOIC_LUT_HOST"] = mode
        res.append(t(lambda: ZoicCamera(0).update(**p)))
    os.environ.pop("ZOIC_LUT_HOST", None)
    print("%s node_update: LUT on the GPU %.1f ms, GPU traces + host draws/replay %.1f ms, host %.1f ms" % ((cfg,) + tuple(res)))
for size in (256, 1024, 2048, 4096):
    img = hexagon_bokeh(size)
    def run():
        c = ZoicCamera(0); c.set_bokeh_image(img); c.update(lensModel=THINLENS, useImage=True, bokehPath="mem:%d" % size)
    print("bokeh %4d^2: update (CDF build + upload) %.1f ms" % (size, t(run, 2)))
