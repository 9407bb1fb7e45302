#!/usr/bin/env python3
"""Algorithmic FLOP per ray of each Kolb config, re-measured with the oracle's own counters (~106 FLOP per
interface visit of traceThroughLensElements + ~130 FLOP per try for the sample / LUT transform / set-up; SURVEY's figures
came from a 480x270x4 probe of the true reference).  Sample: 4096 runs of 256 consecutive rays at tile offsets spread evenly
over the FULL frame of the config (1 M rays), per-ray retry streams.  Writes profiles/flop_model_r04.json, which bench.py reads
for roofline.flop_frac.  CPU only (the oracle is the instrument here, not the product).

    python tools/flop_model.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count, ray_rng_states, synthetic_samples  # noqa: E402

FLOP_PER_VISIT, FLOP_PER_TRY = 106.0, 130.0

out = {"model": "flop/ray = 106 x interface visits/ray + 130 x tries/ray (SURVEY 8d), visits and tries counted by the oracle on 4096 x 256-ray runs spread over the full frame"}
for cfg in ("C2", "C3", "C4", "C5"):
    c = CONFIGS[cfg]
    oc = oracle.OracleCamera()
    if c["bokeh"]:
        oc.set_bokeh_image(hexagon_bokeh())
    oc.update(**camera_params(cfg))
    total = ray_count(cfg)
    runs, run = 4096, 256
    tiles = total // run
    visits = tries = n = zero = retried = 0
    for k in range(0, runs, 256):     # 256 runs per oracle call
        s_parts, st_parts = [], []
        for r in range(k, min(k + 256, runs)):
            base = (tiles * r // runs) * run
            s_parts.append(synthetic_samples(run, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base))
            st_parts.append(ray_rng_states(run, seed=1, ray_index_base=base))
        s, st = np.concatenate(s_parts), np.concatenate(st_parts)
        v0 = oc.surface_visits()
        res = oc.create_rays(s, rng_states=st, threads=8)
        visits += oc.surface_visits() - v0
        tries += int(res["tries"].astype(np.int64).sum()) + len(s)   # the first try counts
        zero += int((res["weight"] == 0).sum())
        retried += int((res["flags"] & 1).sum())
        n += len(s)
    out[cfg] = {"rays": n, "visits_per_ray": round(visits / n, 3), "tries_per_ray": round(tries / n, 3),
                "flop_per_ray": round(FLOP_PER_VISIT * visits / n + FLOP_PER_TRY * tries / n, 1),
                "zero_weight": round(zero / n, 4), "retried": round(retried / n, 4)}
    print(cfg, out[cfg], flush=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "flop_model_r04.json"), "w"), indent=1)
