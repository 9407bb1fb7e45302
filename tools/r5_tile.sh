#!/bin/bash
# tools/r5_tile.sh [tag] -- tile server: parity tests, the per-sample tests that share its kernel, latency table
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-x}
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
echo "=== tile tests"; timeout -k 10 600 python -m pytest tests/test_tile_gpu.py -x -q 2>&1 | tail -15
echo "=== boundary tests"; timeout -k 10 900 python -m pytest tests/test_boundary_gpu.py -x -q 2>&1 | tail -5
echo "=== tile latency"
for args in "1 64 2000 1 1 0" "1 4096 2000 1 1 0" "1 4096 2000 0 1 0" "1 65536 200 1 1 0" "16 65536 100 1 1 0" "16 4096 1000 1 1 0" "64 65536 30 1 1 0" "1 4096 500 1 1 1" "16 65536 40 1 1 1" "1 4096 500 1 1 3" "16 65536 40 1 1 3" "1 4096 2000 1 0 0" "1 64 2000 1 0 0"; do
  timeout -k 5 120 tools/native/tile_latency $LENS $args 2>&1 | tail -1
done | tee gpurun_out/tile_latency_$TAG.txt
for t in tessar_f2.8 fisheye_muller_f4.0 petzval_f1.25; do echo $t; timeout -k 5 120 tools/native/tile_latency zoic_amd/lenses/$t.dat 1 4096 1000 1 1 0 | tail -1; done | tee -a gpurun_out/tile_latency_$TAG.txt
timeout 120 tools/native/sample_latency $LENS 1 20000 1 1 | tail -1
timeout 120 tools/native/sample_latency $LENS 16 20000 1 1 | tail -1
