#!/bin/bash
# tools/r5_shortwaves.sh (needs the build of commit d9afcf9) -- the listed kernel's short-list path at W = 4 / 8 / 16 / 32 rays per wave (ZOIC_SHORT_WAVE_RAYS) against the old path
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() {  # label
  for c in C2 C3 C5; do
    rm -rf gpurun_out/qs_$c
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qs_$c -- python bench.py --only-headline --config $c --steps 10 --warmup 2 > /dev/null 2>&1
    python - <<PY
import csv,glob
for f in glob.glob("gpurun_out/qs_$c/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "kolb" in r["Name"] and int(r["Calls"]) > 2: print("$1 $c", r["Name"][11:40], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
  done
}
for w in 4 8 16 32; do export ZOIC_SHORT_WAVE_RAYS=$w; run W=$w; done
unset ZOIC_SHORT_WAVE_RAYS; run W=auto
export ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_oldshort.so; run old
