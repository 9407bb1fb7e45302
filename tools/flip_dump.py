"""Which rays does FAST decide differently from STRICT?  Runs both kernels on the same slab and saves the samples of the
flipped rays (and a random control set) for offline analysis against the oracle's per-interface margins.
    python tools/flip_dump.py OUTDIR [n] [unchecked]     (unchecked: FAST without its decision check -- the raw FAST/STRICT disagreements)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from zoic_amd import PRECISION_FAST, PRECISION_FAST_UNCHECKED, PRECISION_STRICT, ZoicCamera
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh

out = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 23
FAST_MODE = PRECISION_FAST_UNCHECKED if (len(sys.argv) > 3 and sys.argv[3] == "unchecked") else PRECISION_FAST
os.makedirs(out, exist_ok=True)
for cfg, where in [("C2", 0.5), ("C3", 0.5), ("C4", 0.3), ("C5", 0.5)]:
    c = CONFIGS[cfg]
    cam = ZoicCamera(0)
    if c["bokeh"]:
        cam.set_bokeh_image(hexagon_bokeh())
    cam.update(**camera_params(cfg))
    base = int(c["width"] * int(c["height"] * where)) * c["spp"]
    s = cam.generate_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base)
    res = {}
    for name, mode in (("strict", PRECISION_STRICT), ("fast", FAST_MODE)):
        cam.set_precision(mode)
        res[name] = cam.create_rays(s, ray_index_base=base)["rays"].clone()
    torch.cuda.synchronize()
    fs, ff = res["strict"][:, 7].view(torch.int32), res["fast"][:, 7].view(torch.int32)
    flip = (fs != ff).nonzero().flatten().cpu().numpy()
    print(cfg, "n", n, "flips", len(flip), "frac %.3g" % (len(flip) / n), flush=True)
    sn = s.cpu().numpy()
    np.savez_compressed(os.path.join(out, "flips_%s.npz" % cfg), base=base, idx=flip, samples=sn[flip],
                        strict=res["strict"][flip].cpu().numpy(), fast=res["fast"][flip].cpu().numpy())
    cam.close()
