#!/bin/bash
# tools/r5_first.sh -- round 5, first GPU call: the new tile tests, the whole GPU suite, tile latencies, A/B of the transposed store
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
echo "=== tile tests"; timeout -k 10 600 python -m pytest tests/test_tile_gpu.py -x -q 2>&1 | tail -15
echo "=== tile latency"
for args in "1 4096 2000 1 1 0" "1 4096 2000 0 1 0" "1 64 2000 1 1 0" "1 65536 200 1 1 0" "16 65536 100 1 1 0" "16 4096 1000 1 1 0" "1 4096 500 1 1 1" "16 65536 40 1 1 1" "1 4096 500 1 1 2" "16 65536 40 1 1 2" "1 4096 500 1 1 3" "16 65536 40 1 1 3" "1 4096 2000 1 0 0"; do
  timeout -k 5 120 tools/native/tile_latency $LENS $args 2>&1 | tail -1
done | tee gpurun_out/tile_latency_r05a.txt
echo "=== GPU suite"; timeout -k 10 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
echo "=== A/B transposed store"
REPS=2 NOTEST=1 LIBS="default tr" MODES=fast,unchecked FLIPS=0 bash tools/r3_ab.sh 2>&1 | tee gpurun_out/ab_tr.log
