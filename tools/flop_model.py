#!/usr/bin/env python3
"""Algorithmic FLOP per ray of each Kolb config, measured with the oracle's own counters (~106 FLOP per interface visit of
traceThroughLensElements + ~130 FLOP per try for the sample / LUT transform / set-up), in TWO readings:

  as_written  every interface visit and every try the REFERENCE's loop runs (zoic.cpp:1927-1947) -- what SURVEY 8(d) prices;
  executed    the work the kernels cannot avoid: the same, minus the tries they PROVE away instead of running:
                * a dead pixel (outside the image circle, zero LUT entries) whose first try fails: its 26 retries repeat that try bit
                  for bit -- one try, that try's visits;
                * a retry-dead ray (no retry can reach the rear element, KolbTable::retry*) whose first try fails: its 26 retries die
                  at interface 0 -- the first try and its visits, plus one try's sample / transform for the last draw's state.
              This is the model bench.py's roofline block prices the VALU bound with: with `as_written` a kernel that skips four fifths
              of C5's tries shows 248 % of the FP32 peak -- a fraction above 1 says the model is wrong, not the machine.

Sample: 4096 runs of 256 consecutive rays at tile offsets spread evenly over the FULL frame of the config (1 M rays), per-ray retry
streams.  The dead-pixel / retry-dead classes are taken per ray from the product's own table constants (csrc/lens_system.cpp fill_table,
restated below in numpy: a tables-only camera gives the LUT and the rear element) and each class goes through the oracle on its own, so
that the oracle's visit counter splits by class.  Writes profiles/flop_model_r06.json.  CPU only (the oracle is the instrument here,
not the product).

    python tools/flop_model.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from zoic_amd import ZoicCamera  # noqa: E402
from zoic_amd.workloads import CONFIGS, camera_params, hexagon_bokeh, ray_count, ray_rng_states, synthetic_samples  # noqa: E402

FLOP_PER_VISIT, FLOP_PER_TRY = 106.0, 130.0


def classify(info, lut_boxes, samples, half_sensor):
    """(dead pixel, retry-dead) per sample: kolb_pool_body.hpp setup_ray's two tests with lens_system.cpp fill_table's constants
    (true sin / cos instead of the parabola pair: the classes differ on a vanishing share of the rays, this is a FLOP model)."""
    el = np.asarray(info["elements"], np.float64)
    R0, th0, ap0, cen0 = el[0][0], el[0][1], el[0][3], el[0][4]
    boxes = np.asarray(lut_boxes, np.float64).reshape(-1, 4)
    o0x, o0y = samples[:, 0].astype(np.float64) * half_sensor, samples[:, 1].astype(np.float64) * half_sensor
    dist = np.hypot(o0x, o0y)
    cx = (boxes[:, 0] + boxes[:, 2]) * 0.5
    cy = (boxes[:, 1] + boxes[:, 3]) * 0.5
    ms = np.maximum(np.abs(boxes[:, 2] - cx), np.abs(boxes[:, 3] - cy))
    sc = dist * 8.0
    in_lut = sc <= len(boxes) - 1
    low = np.clip(np.ceil(sc).astype(int), 1, len(boxes) - 1)
    pct = (dist - low * 0.125) * -8.0
    max_scale = np.where(in_lut, (ms[low] + pct * (ms[low - 1] - ms[low])) * 1.05, 0.0)
    tr = np.where(in_lut, cx[low] + pct * (cx[low - 1] - cx[low]), 0.0)
    dead = (max_scale == 0) & (tr == 0) & (o0x != 0) & (o0y != 0)
    retry_dead = np.zeros(len(samples), bool)
    a, dir_z, oz = ap0 * 0.5, -th0, float(info["originShift"])
    if a < abs(R0) and dir_z > 0:
        sag = abs(R0) - np.sqrt(R0 * R0 - a * a)
        zv = cen0 + R0
        zrim = zv - (-1.0 if R0 < 0 else 1.0) * sag
        l1, l2 = (zv - oz) / dir_z, (zrim - oz) / dir_z
        lo, hi = min(l1, l2), max(l1, l2)
        if lo > 1e-3 and np.isfinite(hi):
            k1, rho0, spread = 1.0 - 0.5 * (1 / lo + 1 / hi), a / lo, 0.5 * (1 / lo - 1 / hi)
            max_d = 0.99 * dir_z * np.sqrt(R0 * R0 - a * a) / a
            k = 1.0011 * 1.0011 + 1e-4
            th = np.arctan2(o0y, o0x)
            sn, cs = np.sin(th), np.cos(th)
            ccx, ccy = tr * (cs - sn) - o0x * k1, tr * (sn + cs) - o0y * k1
            reach = (rho0 + dist * spread + np.abs(max_scale) * k) * 1.01 + 1e-4
            dxy = np.abs(max_scale) * k + np.abs(tr) * 1.4158 + dist
            retry_dead = (~dead) & (ccx * ccx + ccy * ccy > reach * reach) & (dxy <= max_d)
    return dead, retry_dead


def main():
    out = {"model": "flop/ray = 106 x interface visits/ray + 130 x tries/ray (SURVEY 8d), visits and tries counted by the oracle on 4096 x 256-ray "
                    "runs spread over the full frame; 'executed' leaves out the tries the kernels prove away (dead pixels, retry-dead rays): tools/flop_model.py"}
    for cfg in ("C2", "C3", "C4", "C5"):
        c = CONFIGS[cfg]
        oc = oracle.OracleCamera()
        pc = ZoicCamera(device=-1)
        if c["bokeh"]:
            oc.set_bokeh_image(hexagon_bokeh())
            pc.set_bokeh_image(hexagon_bokeh())
        p = camera_params(cfg)
        oc.update(**p)
        pc.update(**p)
        info = pc.info()
        _keys, boxes = oc.lut()
        total = ray_count(cfg)
        runs, run = 4096, 256
        tiles = total // run
        acc = dict(n=0, visits=0, tries=0, zero=0, retried=0, ex_visits=0.0, ex_tries=0.0, dead_short=0, retry_dead_short=0)
        for k in range(0, runs, 256):     # 256 runs per oracle call
            s_parts, st_parts = [], []
            for r in range(k, min(k + 256, runs)):
                base = (tiles * r // runs) * run
                s_parts.append(synthetic_samples(run, c["width"], c["height"], c["spp"], seed=1, ray_index_base=base))
                st_parts.append(ray_rng_states(run, seed=1, ray_index_base=base))
            s, st = np.concatenate(s_parts), np.concatenate(st_parts)
            v0 = oc.surface_visits()
            res = oc.create_rays(s, rng_states=st, threads=8)
            acc["visits"] += oc.surface_visits() - v0
            tries = res["tries"].astype(np.int64)
            acc["tries"] += int(tries.sum()) + len(s)   # the first try counts
            acc["zero"] += int((res["weight"] == 0).sum())
            acc["retried"] += int((res["flags"] & 1).sum())
            acc["n"] += len(s)
            # the kernels' shortcuts: rays of a shortcut class that ran out of tries (their first try failed, as every retry did)
            dead, rdead = classify(info, boxes, s, float(p["sensorWidth"]) * 0.5)
            out_of_tries = tries > 25
            groups = {"dp": dead & out_of_tries, "rd": rdead & out_of_tries}
            groups["rest"] = ~(groups["dp"] | groups["rd"])
            gv = {}
            for name, m in groups.items():
                if not m.any():
                    gv[name] = 0
                    continue
                v1 = oc.surface_visits()
                oc.create_rays(s[m], rng_states=st[m], threads=8)
                gv[name] = oc.surface_visits() - v1
            n_dp, n_rd = int(groups["dp"].sum()), int(groups["rd"].sum())
            acc["dead_short"] += n_dp
            acc["retry_dead_short"] += n_rd
            # rest: as written.  dead pixel: 27 identical tries -> one.  retry-dead: first try + 26 one-visit retries -> the first try, and
            # one more try's sample / transform for the last draw's state (finish_dead_ray)
            acc["ex_visits"] += gv["rest"] + gv["dp"] / 27.0 + (gv["rd"] - 26.0 * n_rd)
            acc["ex_tries"] += int(tries[groups["rest"]].sum()) + int(groups["rest"].sum()) + n_dp + 2 * n_rd
        n = acc["n"]
        out[cfg] = {"rays": n, "visits_per_ray": round(acc["visits"] / n, 3), "tries_per_ray": round(acc["tries"] / n, 3),
                    "flop_per_ray": round(FLOP_PER_VISIT * acc["visits"] / n + FLOP_PER_TRY * acc["tries"] / n, 1),
                    "executed_visits_per_ray": round(acc["ex_visits"] / n, 3), "executed_tries_per_ray": round(acc["ex_tries"] / n, 3),
                    "executed_flop_per_ray": round(FLOP_PER_VISIT * acc["ex_visits"] / n + FLOP_PER_TRY * acc["ex_tries"] / n, 1),
                    "dead_pixel_shortcut": round(acc["dead_short"] / n, 4), "retry_dead_shortcut": round(acc["retry_dead_short"] / n, 4),
                    "zero_weight": round(acc["zero"] / n, 4), "retried": round(acc["retried"] / n, 4)}
        print(cfg, out[cfg], flush=True)
    json.dump(out, open(os.path.join(ROOT, "profiles", "flop_model_r06.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
