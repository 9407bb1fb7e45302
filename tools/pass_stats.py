"""Pass statistics of the Kolb kernels (debug build: tools/build_variant.sh stats -DZOIC_PASS_STATS).
    ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_stats.so python tools/pass_stats.py [--configs C2,C3,C4,C5] [--modes fast]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicCamera, _capi
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count
MODES = {'fast': PRECISION_FAST, 'unchecked': PRECISION_FAST_UNCHECKED, 'strict': PRECISION_STRICT}
ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3,C4,C5")
ap.add_argument("--modes", default="unchecked")
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
    for mode in a.modes.split(","):
        cam.set_precision(MODES[mode])
        out = (ctypes.c_ulonglong * 8)()
        cam.create_rays(s); torch.cuda.synchronize()
        lib.zoic_debug_pass_stats(out, 1)
        lib.zoic_debug_region_cycles((ctypes.c_ulonglong * 16)(), 1)
        cam.create_rays(s); torch.cuda.synchronize()
        lib.zoic_debug_pass_stats(out, 1)
        A, B, it, look, tr, candl, act, fin = [int(v) for v in out]
        rc = (ctypes.c_ulonglong * 16)()
        lib.zoic_debug_region_cycles(rc, 1)
        tot = float(sum(rc)) or 1.0
        names = ["top/flush", "setup_ray", "first try", "pool pop", "retry search", "advance", "trace", "finish+store", "push", "tail"]
        print("   tail: %.1f %% of the wave time is passes without fresh work (drain)" % (100.0 * rc[9] / max(tot - rc[9], 1.0)))
        tot -= rc[9]
        print("   wave-time shares: " + ", ".join("%s %.1f%%" % (names[i], 100.0 * rc[i] / tot) for i in range(9)))
        print("%s %-9s rays %d: per 64 rays: A %.3f B %.3f passes, search iterations %.3f (looking %.1f lanes each), traces %.3f (cand %.1f lanes each), active lanes/pass %.1f, finished %d" % (
            cfg, mode, n, A * 64 / n, B * 64 / n, it * 64 / n, look / max(it, 1), tr * 64 / n, candl / max(tr, 1), act / max(A + B, 1), fin))
    cam.close()
