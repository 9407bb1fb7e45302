"""tools/soak.py [launches]: repeat-launch soak of the persistent kernels: every launch of a config must reproduce the
first one bit for bit (records parked in LDS, work cursors, look for races / lost rays), no launch may hang."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import ZoicCamera, PRECISION_FAST, PRECISION_STRICT
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
cases = [("C3", PRECISION_FAST, {}, 1 << 24), ("C3", PRECISION_STRICT, {}, 1 << 23), ("C2", PRECISION_FAST, {}, 0), ("C4", PRECISION_FAST, {}, 1 << 25),
         ("C1", PRECISION_FAST, dict(opticalVignettingDistance=4.0), 1 << 24), ("C1", PRECISION_STRICT, dict(opticalVignettingDistance=2.0, useImage=True, bokehPath="procedural:hexagon256"), 1 << 23),
         ("C5", PRECISION_FAST, {}, 1 << 26)]
for cfg, prec, extra, n in cases:
    c = CONFIGS[cfg]
    p = dict(camera_params(cfg), **extra)
    cam = ZoicCamera(0)
    if p.get("useImage"):
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**p); cam.set_precision(prec)
    n = n or c["width"] * c["height"] * c["spp"]
    base = c["width"] * (c["height"] // 3) * c["spp"]
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=5, ray_index_base=base)
    first = cam.create_rays(s, ray_index_base=base)["rays"].clone()
    out = dict(rays=torch.empty_like(first))
    torch.cuda.synchronize(); t0 = time.perf_counter(); bad = 0
    for i in range(launches):
        out["rays"].fill_(float("nan"))
        cam.create_rays(s, ray_index_base=base, out=out)
        if i % 7 == 0 or i == launches - 1:
            if not torch.equal(out["rays"].view(torch.int32), first.view(torch.int32)):
                bad += 1
    torch.cuda.synchronize()
    cnt = cam.counters()
    ok = bad == 0 and cnt["succesRays"] + cnt["vignettedRays"] == (launches + 1) * n
    print("%s %-6s %s n=%d: %d launches in %.1f s, mismatching launches %d, counters %s -> %s" % (
        cfg, "fast" if prec == PRECISION_FAST else "strict", extra or "", n, launches, time.perf_counter() - t0, bad, "ok" if cnt["succesRays"] + cnt["vignettedRays"] == (launches + 1) * n else "WRONG", "OK" if ok else "FAIL"))
    assert ok
print("soak passed")
