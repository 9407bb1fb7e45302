"""Inner-loop kernel benchmark: kernel ms / Grays/s of the production kernels per config, and (--flips) the decision
flips of FAST against STRICT on the GPU itself (STRICT is bit-exact against the oracle, tests/test_parity_gpu.py).
    ZOIC_AMD_LIB=path/to/variant.so python tools/kbench.py [--configs C2,C3,C4,C5] [--modes fast] [--steps 10] [--flips N] [--tag name]
"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicCamera
MODES = {'fast': PRECISION_FAST, 'unchecked': PRECISION_FAST_UNCHECKED, 'strict': PRECISION_STRICT}
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="C2,C3,C4,C5")
ap.add_argument("--modes", default="fast")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--flips", type=int, default=0, help="samples per config for the fast-vs-strict decision comparison")
ap.add_argument("--rays", type=int, default=0)
ap.add_argument("--flipmode", default="fast", help="the mode compared against strict by --flips: fast | unchecked")
ap.add_argument("--tag", default=os.path.basename(os.environ.get("ZOIC_AMD_LIB", "default")))
a = ap.parse_args()
dev = torch.device("cuda", 0)
row = {"tag": a.tag}
for cfg in a.configs.split(","):
    c = CONFIGS[cfg]
    cam = ZoicCamera(0)
    if c["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg))
    n = a.rays or ray_count(cfg)
    if cfg == "C5" and not a.rays:
        n //= 4          # a quarter frame (rows 0..1079 of 4320): same kernel, shorter run
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1)
    out = dict(rays=torch.empty((n, 8), dtype=torch.float32, device=dev))
    for mode in a.modes.split(","):
        cam.set_precision(MODES[mode])
        for _ in range(2):
            cam.create_rays(s, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            cam.create_rays(s, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        row["%s_%s" % (cfg, mode)] = round(n / ms / 1e6, 2)
        print("%-22s %s %-6s n=%-11d %8.4f ms  %7.2f Grays/s" % (a.tag, cfg, mode, n, ms, n / ms / 1e6), flush=True)
    if a.flips:
        m = min(a.flips, n)
        base = int(c["width"] * int(c["height"] * (0.3 if cfg == "C4" else 0.5))) * c["spp"]
        s2 = cam.generate_samples(m, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
        res = {}
        for name, mode in (("strict", PRECISION_STRICT), ("fast", MODES[a.flipmode])):
            cam.set_precision(mode)
            cam.reset_counters()
            res[name] = cam.create_rays(s2, ray_index_base=base)["rays"].clone()
            res[name + "_c"] = cam.counters()
        fs, ff = res["strict"][:, 7].view(torch.int32), res["fast"][:, 7].view(torch.int32)
        same = fs == ff
        live = same & (res["strict"][:, 6] != 0)
        dd = (res["strict"][live, 3:6].double() - res["fast"][live, 3:6].double())
        rmse = float(dd.pow(2).sum(1).mean().sqrt()) if live.any() else 0.0
        nf = int((~same).sum())
        row["%s_flips" % cfg] = nf
        print("%-22s %s flips %d of %d (%.3g)  dir rmse %.3g  counters equal %s" % (a.tag, cfg, nf, m, nf / m, rmse, res["strict_c"] == res["fast_c"]), flush=True)
    cam.close()
    del s, out
    torch.cuda.empty_cache()
print("KBENCH " + json.dumps(row))
