#!/usr/bin/env python3
"""Markdown rows of DESIGN.md section 5 from the committed profiles of a round: python tools/roofline_table.py r04"""
import json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zoic_amd.workloads import ray_count
rows = [("C3", "fast", "**C3 fast (decision-safe) -- headline**"), ("C3", "unchecked", "C3 unchecked"), ("C3", "strict", "C3 strict (bit-exact)"),
        ("C2", "fast", "C2 fast"), ("C4", "fast", "C4 fast"), ("C5", "fast", "C5 fast"), ("C1", "fast", "C1 thin lens")]
flop = json.load(open(os.path.join(ROOT, "profiles", "flop_model_r06.json")))
print("| config (mode) | Grays/s | ms/frame | kernel medians us | lane-instr/ray | trans/ray | issue slots / instr | VALU issue of 1.229 T/s | VALU pipe by mix | lane util | waves waiting | avg waves/SIMD | HBM at 44 B/ray of 8 TB/s | executed FLOP of 157 TF (as written) | bound | PMC traffic / 44 B-algorithmic |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for cfg, mode, label in rows:
    d = os.path.join(ROOT, "profiles", "%s_%s_%s" % (R, mode, cfg))
    try:
        s = json.load(open(os.path.join(d, "summary.json")))
        b = json.load(open(os.path.join(d, "bench_line.json")))
    except Exception as e:
        print("| %s | missing: %s |" % (label, e)); continue
    n = ray_count(cfg)
    disp = s["dispatch_us"] if isinstance(s["dispatch_us"], dict) else eval(s["dispatch_us"])
    meds = " + ".join("%.0f" % v["median_us"] for k, v in disp.items() if v["dispatches"] >= 10)
    sec = b["ms_per_step"] * 1e-3
    f = lambda k: float(s[k]) if s.get(k) not in (None, "nan") else float("nan")
    fl, flx = flop.get(cfg, {}).get("flop_per_ray"), flop.get(cfg, {}).get("executed_flop_per_ray")
    hbm = 44 * n / sec / 8e12
    bound = "HBM" if (not flx or 157.3e12 / flx > 8e12 / 44) else "VALU"
    print("| %s | %.1f | %.3f | %s | %.0f | %.1f | %.2f | %.2f | %.2f | %.2f | %.0f %% | %.1f | %.1f %% | %s | %s | %.2f / %.2f GB |" % (
        label, b["value"] / 1e3, b["ms_per_step"], meds, f("lane_instr_per_ray"), f("trans_per_ray"), f("valu_issue_slots_per_instr"), f("valu_issue_frac_2cyc"),
        f("valu_pipe_frac_by_mix"), f("valu_thread_util"),
        100 * f("wave_cycles_wait_any"), f("avg_waves_per_simd"), 100 * hbm,
        ("%.0f %% (%.0f %%)" % (100 * flx * n / sec / 157.3e12, 100 * fl * n / sec / 157.3e12)) if fl else "--", bound, f("hbm_bytes_per_launch") / 1e9, 44 * n / 1e9))
