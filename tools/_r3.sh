cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ZOIC_GUARD_SCALE=100000 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_all -- python tools/kbench.py --configs C3 --modes fast,strict --steps 3 > gpurun_out/prof_all.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob("gpurun_out/prof_all/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "refill" in r["Name"]: print("%-58s calls %5s avg_us %10.1f" % (r["Name"][11:69], r["Calls"], float(r["AverageNs"])/1e3))
PY
