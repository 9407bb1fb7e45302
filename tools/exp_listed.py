"""Developer probe of the listed kernel (kolb_listed_body.hpp): (1) a ray's bits must not depend on the length of the work list
it was on -- 16.6 M rays of a config in ONE launch (long list: batches + pool) against the same rays in 1 M-ray launches (short
lists: listed_short_hybrid) and against the per-sample kernel on a few hundred listed rays; (2) decision flips against the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoic_amd import ZoicCamera, PRECISION_FAST
from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples, hexagon_bokeh
import oracle

cfgs = sys.argv[1:] or ["C4", "C2", "C3", "C5"]
for cfg in cfgs:
    c = CONFIGS[cfg]
    cam = ZoicCamera(0)
    if c["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg))
    cam.set_precision(PRECISION_FAST)
    n, base = 1 << 24, (c["width"] * (c["height"] // 3)) * c["spp"]
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    one = cam.create_rays(s, ray_index_base=base)["rays"].clone()
    parts = torch.empty_like(one)
    step = 1 << 20
    for a in range(0, n, step):
        parts[a:a + step] = cam.create_rays(s[a:a + step], ray_index_base=base + a)["rays"]
    torch.cuda.synchronize()
    same = torch.equal(one.view(torch.int32), parts.view(torch.int32))
    nbad = int((one.view(torch.int32) != parts.view(torch.int32)).any(1).sum())
    print("%s: one 16.8 M launch == sixteen 1 M launches: %s (%d rays differ)" % (cfg, same, nbad), flush=True)
    # against the oracle on the first 1 M rays
    m = 1 << 20
    oc = oracle.OracleCamera()
    if c["bokeh"]:
        oc.set_bokeh_image(hexagon_bokeh())
    oc.update(**camera_params(cfg))
    sh = s[:m].cpu().numpy()
    ref = oc.create_rays(sh, rng_states=ray_rng_states(m, seed=1, ray_index_base=base), threads=16)
    got = one[:m].cpu().numpy()
    flags = got[:, 7].view(np.uint32).astype(np.uint8)
    flips = int((flags != ref["flags"]).sum())
    live = (flags == ref["flags"]) & (ref["weight"] != 0)
    dd = got[live, 3:6].astype(np.float64) - ref["dir"][:, live].T.astype(np.float64)
    print("   vs oracle, 1 M rays: flips %d, dir RMSE %.3g, weight equal %s" % (flips, np.sqrt((dd ** 2).sum(1).mean()), bool(np.array_equal(got[:, 6], ref["weight"]))), flush=True)
    cam.close()
