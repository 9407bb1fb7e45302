#!/bin/bash
# tools/pmc_extra.sh <tag> [bench args]: extra SQ PMC passes (issue mix, ifetch, branches) for bottleneck hunting; run on the GPU box.
# TA_*/TCP_* counter groups hung rocprofv3 on this pool (a 15-minute timeout in round 1): not collected; every pass runs under `timeout`.
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmcx_$TAG
mkdir -p $OUT
BENCH="python bench.py --steps 3 --warmup 1 --only-headline $*"
i=0
for grp in "SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed: $grp" >> $OUT/errors.txt
done
python - $OUT <<'PY'
import csv, glob, sys, collections
tot = {}
for d in sorted(glob.glob(sys.argv[1] + "/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "kolb" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        tot[k] = sum(v) / len(v)
for k in sorted(tot): print("%-44s %.4g" % (k, tot[k]))
PY
cat $OUT/errors.txt 2>/dev/null
