"""Developer probe: strict/fast GPU output vs the oracle on small slabs of every config (prints, no asserts)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zoic_amd import ZoicCamera, PRECISION_FAST, PRECISION_STRICT
from zoic_amd.workloads import CONFIGS, camera_params, ray_rng_states, synthetic_samples, hexagon_bokeh
import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
img = hexagon_bokeh()
for cfg in ("C1", "C2", "C3", "C4", "C5"):
    c = CONFIGS[cfg]
    p = camera_params(cfg)
    samples = synthetic_samples(n, c["width"], c["height"], c["spp"], seed=1, ray_index_base=12345 * c["spp"])
    states = ray_rng_states(n, seed=1, ray_index_base=12345 * c["spp"])
    cam = ZoicCamera(0); oc = oracle.OracleCamera()
    if c["bokeh"]:
        cam.set_bokeh_image(img); oc.set_bokeh_image(img)
    t = time.time(); cam.update(**p); tu = time.time() - t
    oc.update(**p)
    t = time.time(); ref = oc.create_rays(samples, rng_states=states); tc = time.time() - t
    for mode, name in ((PRECISION_STRICT, "strict"), (PRECISION_FAST, "fast")):
        if cfg == "C1" and name == "fast":
            continue
        cam.set_precision(mode)
        got = cam.create_rays(samples, ray_index_base=12345 * c["spp"])
        same_flags = np.array_equal(got["flags"], ref["flags"])
        bits = np.array_equal(got["planes"].view(np.uint32), ref["planes"].view(np.uint32))
        agree = (got["flags"] == ref["flags"]) & (ref["weight"] != 0)
        dd = got["dir"][:, agree] - ref["dir"][:, agree]
        rmse = float(np.sqrt((dd.astype(np.float64) ** 2).sum(0).mean())) if agree.any() else 0.0
        do = got["origin"][:, agree] - ref["origin"][:, agree]
        ormse = float(np.sqrt((do.astype(np.float64) ** 2).sum(0).mean())) if agree.any() else 0.0
        nbad = int((got["planes"].view(np.uint32) != ref["planes"].view(np.uint32)).any(0).sum())
        print("%s %-6s n=%d flags_equal=%s bit_exact=%s mismatching_rays=%d flip_frac=%.3g dirRMSE=%.3g originRMSE=%.3g zero_w=%.3f retried=%.3f | update %.2fs oracle %.2f Mrays/s" % (
            cfg, name, n, same_flags, bits, nbad, float((got["flags"] != ref["flags"]).mean()), rmse, ormse,
            float((ref["weight"] == 0).mean()), float((ref["flags"] & 1).mean()), tu, n / tc / 1e6), flush=True)
    print("   counters gpu", cam.counters(), "oracle", oc.counters(), flush=True)
    cam.close()
