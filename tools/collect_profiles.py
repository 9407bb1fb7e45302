"""Copy the judged summaries of a tools/profile.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
kernel_stats.csv (rocprofv3 --kernel-trace --stats), the PMC counter rows of the dominant kernel, summary.json, and the
HBM traffic entry bench.py reads (profiles/pmc_traffic.json)."""
import csv, glob, json, os, shutil, subprocess, sys
tag, nrays, key = sys.argv[1], int(sys.argv[2]), sys.argv[3]      # e.g. r01_fast_v5_C3 132710400 C3_fast
src, dst = "gpurun_out/prof_" + tag, "profiles/" + tag
os.makedirs(dst, exist_ok=True)
for f in glob.glob(src + "/trace/*/*_kernel_stats.csv"):
    shutil.copy(f, dst + "/kernel_stats.csv")
rows = []
for f in sorted(glob.glob(src + "/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "kolb" in r["Kernel_Name"] or "thin_rays" in r["Kernel_Name"]:
            rows.append({k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                           "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp")})
with open(dst + "/pmc_counters.csv", "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
out = subprocess.check_output([sys.executable, "tools/pmc_summary.py", src, str(nrays)] + (["thin_r"] if ("thin" in tag or key.startswith("C1")) else []), text=True)
open(dst + "/summary.txt", "w").write(out)
shutil.copy(src + "/summary.json", dst + "/summary.json")
if os.path.exists(src + "/bench_line.json"):
    shutil.copy(src + "/bench_line.json", dst + "/bench_line.json")
s = json.load(open(dst + "/summary.json"))
tp = "profiles/pmc_traffic.json"
t = json.load(open(tp)) if os.path.exists(tp) else {}
sys.path.insert(0, ".")
from bench import csrc_sha16   # the fingerprint of the kernel sources this profile was taken on: bench.py replays the entry only on the same sources
t[key] = {"csrc_sha16": csrc_sha16(), "hbm_bytes_per_launch": s["hbm_bytes_per_launch"], "fetch_bytes_x2_corrected": s["fetch_bytes(x2 corrected)"],
          "write_bytes": s["write_bytes"], "lane_instr_per_ray": s["lane_instr_per_ray"], "valu_thread_util": s["valu_thread_util"],
          "trans_per_ray": s.get("trans_per_ray"), "valu_issue_slots_per_instr": s.get("valu_issue_slots_per_instr"),
          "pipeline_us_per_launch": s.get("pipeline_us_per_launch"), "dispatch_us": s.get("dispatch_us"),
          "source": dst + "/pmc_counters.csv (FETCH_SIZE x2 per MI355X_MICROARCH.md, separate --pmc passes; counters summed over the launch's kernel pipeline)"}
json.dump(t, open(tp, "w"), indent=1)
print(out)
