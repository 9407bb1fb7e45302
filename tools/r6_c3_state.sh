#!/bin/bash
# which state is C3 in on this box (40.6 .. 45.4 Grays/s), and what do the memory-path latency counters say in that state?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3_state.log; : > $O
line() { timeout 200 python bench.py --only-headline --config $1 --steps 20 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for cfg in C3 C4; do
  D=gpurun_out/c3_state_$cfg; rm -rf $D; mkdir -p $D
  echo "== $cfg before: $(line $cfg)" >> $O
  i=0
  for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" \
             "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $D/pmc$i -- python bench.py --only-headline --config $cfg --steps 5 --warmup 5 > $D/pmc$i.log 2>&1 || echo "pass $i failed: $grp" >> $O
  done
  echo "== $cfg after: $(line $cfg)" >> $O
  python tools/r6_c3_state.py $D >> $O 2>&1
  rm -rf $D
done
cat $O
