"""tools/pmc_kernel.py <prof dir> <kernel substring>: PMC counters of one kernel (mean per dispatch) from a tools/profile.sh directory."""
import collections, csv, glob, sys
base, pat = sys.argv[1], sys.argv[2]
tot = {}
for d in sorted(glob.glob(base + "/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            tot["_regs"] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size"))
    for k, v in acc.items():
        tot[k] = sum(v) / len(v)
for k in sorted(tot):
    print("%-28s %s" % (k, tot[k] if k.startswith("_") else "%.5g" % tot[k]))
g = lambda k: tot.get(k, float("nan"))
print("wave-instr VALU per wave: %.0f; util %.2f; wait_inst %.2f; wait_any %.2f; active %.2f" % (
    g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64), g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
    g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
print("kernel cycles (GRBM/8) %.0f; wave-cycles per wave %.0f; avg waves/SIMD %.2f" % (g("GRBM_GUI_ACTIVE") / 8, g("SQ_WAVE_CYCLES") * 4 / g("SQ_WAVES"), g("SQ_WAVE_CYCLES") * 4 / (g("GRBM_GUI_ACTIVE") / 8 * 1024)))
