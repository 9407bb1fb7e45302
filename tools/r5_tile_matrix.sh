#!/bin/bash
# worker-group sweep of the tile server (ZOIC_TILE_WORKER_GROUPS) on a few shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
for g in 16 64 128 256; do
  export ZOIC_TILE_WORKER_GROUPS=$g
  for args in "1 256 2000 1 1 0" "1 1024 2000 1 1 0" "1 4096 2000 1 1 0" "1 16384 500 1 1 0" "1 65536 200 1 1 0" "16 4096 1000 1 1 0" "16 65536 60 1 1 0" "4 65536 100 1 1 0"; do
    echo -n "groups=$g "; timeout -k 5 120 tools/native/tile_latency $LENS $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['threads'], d['samples_per_tile'], 'p50', d['p50_us'], 'p99', d['p99_us'], 'Mrays/s', d['mrays_s'])"
  done
done | tee gpurun_out/tile_matrix_${1:-x}.txt
