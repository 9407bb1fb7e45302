#!/bin/bash
# tools/profile.sh <tag> [bench args...]  -- run on the GPU box (inside gpurun).  Writes gpurun_out/prof_<tag>/...
# 0) the plain bench line of the same command (no profiler): bench_line.json -- the ms_per_step the profile must fit into;
# 1) kernel trace + stats (csv) over ZOIC_PROFILE_STEPS (20) timed + ZOIC_PROFILE_WARMUP (10) warm-up dispatches; 2..n) one PMC pass per counter group (never combined
# with sys/hip traces; 5 dispatches each).  The last group is the VALU instruction classes (SQ_INSTS_VALU_TRANS_F32 is the counter that exists on
# gfx950: round 5 asked for SQ_INSTS_VALU_TRANS and every summary printed trans_per_ray = NaN).
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
S=${ZOIC_PROFILE_STEPS:-20}; W=${ZOIC_PROFILE_WARMUP:-10}
python bench.py --steps $S --warmup $W --only-headline $* 2> $OUT/bench_line.err | tail -1 > $OUT/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps $S --warmup $W --only-headline $* > $OUT/trace.log 2>&1
BENCH="python bench.py --steps 5 --warmup $W --only-headline $*"
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed: $grp" >> $OUT/errors.txt
done
