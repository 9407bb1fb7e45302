#!/bin/bash
# tools/r4_quick.sh [configs...] -- on the GPU box: kernel durations (rocprofv3 --kernel-trace --stats) + the plain bench line per config
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in ${@:-C2 C3 C4 C5}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q_$c -- python bench.py --only-headline --config $c --steps 10 --warmup 2 > /dev/null 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/q_$c/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "kolb" in r["Name"] and int(r["Calls"]) > 2: print("$c", r["Name"][11:60], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
  python bench.py --only-headline --config $c --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c bench', d['value'], d['ms_per_step'])"
done
