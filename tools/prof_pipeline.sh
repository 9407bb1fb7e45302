#!/bin/bash
# tools/prof_pipeline.sh [configs] [mode]: per-kernel split of a Kolb launch (rocprofv3 kernel trace of tools/kbench.py); run on the GPU box
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
C=${1:-C2,C3,C5}; M=${2:-fast}
rm -rf gpurun_out/trace_pipe
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_pipe -- python tools/kbench.py --configs $C --modes $M --steps 10 > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("gpurun_out/trace_pipe/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "kolb" in r["Name"] or "fill" in r["Name"]:
            print("%-72s calls %4s  avg %.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
