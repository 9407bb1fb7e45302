#!/bin/bash
# tools/pmc_extra.sh <config> [precision]: extra PMC groups (memory pipeline, instruction fetch, instruction classes) of the headline kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
C=${1:-C3}; P=${2:-fast}
OUT=gpurun_out/pmcx_${C}_$P
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_BRANCH" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_BUSY_CYCLES SQ_CYCLES SQ_LEVEL_WAVES" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCC_REQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python bench.py --steps 5 --warmup 2 --only-headline --config $C --precision $P > $OUT/pmc$i.log 2>&1 || echo "group $i failed"
done
python - <<PY
import csv, glob, collections
tot = {}
for d in sorted(glob.glob("$OUT/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "kolb_pool" in r["Kernel_Name"] and "listed" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():   # full-size dispatches only (node_update's self-check launches the same kernels over 8192 rays)
        big = [x for x in v if x > 0.5 * max(v)] or v
        tot[k] = sum(big) / len(big)
for k in sorted(tot):
    print("%-36s %.5g" % (k, tot[k]))
PY
