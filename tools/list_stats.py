"""Work-list statistics of decision-safe FAST (debug build: tools/build_variant.sh lstats -DZOIC_PASS_STATS -DZOIC_PS_LIST).
    ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_lstats.so python tools/list_stats.py [--configs C2,C3,C4,C5]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import PRECISION_FAST, ZoicCamera, _capi
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3,C4,C5")
a = ap.parse_args()
lib = _capi.load()
for cfg in a.configs.split(","):
    c = CONFIGS[cfg]
    cam = ZoicCamera(0)
    if c["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg))
    n = min(ray_count(cfg), 1 << 27)
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1)
    cam.set_precision(PRECISION_FAST)
    out = (ctypes.c_ulonglong * 8)()
    cam.create_rays(s); torch.cuda.synchronize()
    lib.zoic_debug_pass_stats(out, 1)
    cam.create_rays(s); torch.cuda.synchronize()
    lib.zoic_debug_pass_stats(out, 1)
    listed, triesAt, fin, triesFin = [int(v) for v in out[:4]]
    print("%s rays %d: listed %d (%.3f%%), tries when listed %.2f on average; the listed kernel finished %d of them with %.2f tries on average"
          % (cfg, n, listed, 100.0 * listed / n, triesAt / max(listed, 1), fin, triesFin / max(fin, 1)))
    cam.close()
