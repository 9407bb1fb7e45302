"""Summarise a tools/profile.sh output directory: kernel stats + PMC-derived rates of one camera_create_ray launch.
A Kolb launch is a short pipeline of kernels (kolb_pool_body.hpp main kernel + kolb_listed_body.hpp listed kernel): counters
are averaged per dispatch for every kernel of the pipeline and then SUMMED over the pipeline, so every figure is per launch."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
base, nrays = sys.argv[1], float(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else "kolb"
st = glob.glob(base + "/trace/*/*_kernel_stats.csv")
launch_us = 0.0
if st:
    for r in list(csv.DictReader(open(st[0]))):
        if pat in r["Name"]:
            print("%-66s calls %s avg_us %.1f" % (r["Name"][:66], r["Calls"], float(r["AverageNs"]) / 1e3))
            launch_us += float(r["AverageNs"]) / 1e3
    print("pipeline per launch (sum of kernel averages): %.1f us" % launch_us)
# per-dispatch durations (kernel trace): median and minimum of every kernel of the launch over the TIMED dispatches (the last 20)
dispatch = {}
for f in glob.glob(base + "/trace/*/*_kernel_trace.csv"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in per.items():
        v = sorted(v[-20:])
        dispatch[k.split("(")[0].replace("void zoic::", "")] = {"dispatches": len(v), "median_us": round(v[len(v) // 2], 1), "min_us": round(v[0], 1), "max_us": round(v[-1], 1)}
        print("%-60s %d timed dispatches: median %.1f min %.1f max %.1f us" % (k[:60], len(v), v[len(v) // 2], v[0], v[-1]))
bench_line = None
try:
    bench_line = json.load(open(base + "/bench_line.json"))
    print("bench line of the same call (no profiler): value %.1f Mrays/s, ms_per_step %.4f, kernel_ms %.4f" % (
        bench_line["value"], bench_line["ms_per_step"], bench_line["roofline"]["kernel_ms"]))
except Exception as e:
    print("no bench line:", e)
tot = {}
for d in sorted(glob.glob(base + "/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))     # counter -> kernel -> values
    for r in csv.DictReader(open(d)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Kernel_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            if "listed" not in r["Kernel_Name"]:
                tot["_vgpr"], tot["_sgpr"], tot["_lds"] = r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size")
    # per kernel: the mean over its FULL-SIZE dispatches only -- node_update's self-check of the fast modes (capi.cpp fast_self_check)
    # launches the same kernels over 8192 probe rays, and an average that includes those dispatches is diluted by 1 / (dispatches)
    # ... and of those only the LAST five: bench.py's timed launches.  The launches before them are warm-up and, since round 6, the
    # placement probe (zoic_amd/placement.py: the same frame on other pairs of buffers, some of them slower -- their wait counters are not the run's)
    def full(v):
        v = [x for _, x in sorted(v)]
        m = max(v)
        big = [x for x in v if x > 0.5 * m] if m > 0 else v
        big = big[-5:]
        return sum(big) / len(big)
    for k, per_kernel in acc.items():
        # kernels that ran once or twice in the whole process are the self-check's, not the launch's
        tot[k] = sum(full(v) for kn, v in per_kernel.items() if len(v) >= 3 or len(per_kernel) == 1)
g = lambda k: tot.get(k, float("nan"))
# registers / spills of the launch's main kernel from the CODE OBJECT (tools/code_object_regs.py), not from the profiler's CSV
# (rocprofv3's VGPR_Count column printed 40 for the 79-VGPR headline kernel)
main_kernel, regs = None, None
try:
    import code_object_regs
    timed = {k: d for k, d in dispatch.items() if d["dispatches"] >= 10}
    if timed:
        main_kernel = max(timed, key=lambda k: timed[k]["median_us"])
        regs = code_object_regs.kernel_resources(os.environ.get("ZOIC_AMD_LIB") or os.path.join(code_object_regs.ROOT, "zoic_amd", "libzoic_amd.so")).get(main_kernel)
except Exception as e:
    print("code object registers unavailable:", e)
# VALU instruction classes (profile.sh's last PMC pass) and the issue-slot model they give: op_rate (profiles/ubench_r01.txt) measures
# f32 fma / mul / add and 32-bit integer add / logic at ~1 wave-instruction per 2 cycles per SIMD ("full rate"), compares, selects, shifts,
# conversions, v_max, lane moves at HALF of it, v_sqrt / v_rsq / v_rcp at a QUARTER.  slots = full + 2 x half + 4 x quarter, in units of one
# full-rate issue; `other` = the VALU instructions in none of the counted classes (compares, selects, moves, lane moves ...), priced at half rate.
cls = {k: g("SQ_INSTS_VALU_" + k) for k in ("FMA_F32", "MUL_F32", "ADD_F32", "INT32", "CVT", "TRANS_F32")}
valu = g("SQ_INSTS_VALU")
counted = sum(cls.values())
other = valu - counted
issue_slots = cls["FMA_F32"] + cls["MUL_F32"] + cls["ADD_F32"] + cls["INT32"] + 2 * (cls["CVT"] + other) + 4 * cls["TRANS_F32"]
cyc = g("GRBM_GUI_ACTIVE") / 8
simd = cyc * 1024
out = {
    "pipeline_us_per_launch": (sum(d["median_us"] for d in dispatch.values() if d["dispatches"] >= 10) or launch_us),
    "dispatch_us": dispatch,
    "bench_line_same_call": {k: bench_line[k] for k in ("value", "ms_per_step")} if bench_line else None,
    "dominant_kernel_median_fits_ms_per_step": (max(d["median_us"] for d in dispatch.values()) <= bench_line["ms_per_step"] * 1e3) if (bench_line and dispatch) else None,
    "main_kernel": main_kernel,
    "code_object (main kernel)": regs,
    "lds_block_bytes (profiler)": tot.get("_lds"),
    "lane_instr_per_ray": g("SQ_INSTS_VALU") * 64 / nrays,
    "valu_thread_util": g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64),
    "salu_per_valu": g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"),
    "wave_cycles_wait_any": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
    "wave_cycles_wait_inst": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "wave_cycles_active": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
    "avg_waves_per_simd": g("SQ_WAVE_CYCLES") * 4 / simd,
    "valu_issue_frac_2cyc": g("SQ_INSTS_VALU") * 2 / simd,
    "valu_busy_quadcycles": g("SQ_ACTIVE_INST_VALU") * 4 / simd,
    "vmem_rd_per_ray": g("SQ_INSTS_VMEM_RD") * 64 / nrays, "vmem_wr_per_ray": g("SQ_INSTS_VMEM_WR") * 64 / nrays,
    "smem_per_wave_instr": g("SQ_INSTS_SMEM") / g("SQ_INSTS_VALU"),
    "lds_instr_per_ray": g("SQ_INSTS_LDS") * 64 / nrays,
    "trans_per_ray": cls["TRANS_F32"] * 64 / nrays,
    "valu_mix_per_ray": {k.lower(): round(v * 64 / nrays, 1) for k, v in cls.items()} | {"other(cmp/select/mov/lane...)": round(other * 64 / nrays, 1)} if counted == counted else None,
    "valu_issue_slots_per_instr": issue_slots / valu,
    "valu_pipe_frac_by_mix": issue_slots * 2 / simd,   # of the SIMDs' cycles, if every full-rate slot takes 2 cycles: ~1.0 = the VALU pipe never idles
    "hbm_bytes_per_launch": (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024,
    "fetch_bytes(x2 corrected)": 2 * g("FETCH_SIZE") * 1024, "write_bytes": g("WRITE_SIZE") * 1024,
    "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
    "gpu_ms(cycles/2.4GHz)": cyc / 2.4e6,
}
for k, v in out.items():
    print("%-28s %s" % (k, ("%.4g" % v) if isinstance(v, float) else v))
json.dump({k: v for k, v in out.items()}, open(base + "/summary.json", "w"), indent=1, default=str)
