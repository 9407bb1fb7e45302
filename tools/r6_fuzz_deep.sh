#!/bin/bash
# tools/r6_fuzz_deep.sh -- the deepest fuzz run on the round's final build (depths of rounds 3-5): parameter cameras 2000 STRICT + 1000 FAST, 3000 machine-made lenses,
# 1500 bokeh images, 600 hostile + 600 wild, 150 update sequences, 300 per-sample cameras, 1500 tile-fuzz cameras.  Then the tile server's soak incl. this round's
# sleeping waits (64 threads) and device-resident requests.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
export ZOIC_FUZZ_EXAMPLES=2000 ZOIC_FUZZ_EXAMPLES_FAST=1000 ZOIC_FUZZ_EXAMPLES_LENS=3000 ZOIC_FUZZ_EXAMPLES_IMAGE=1500 ZOIC_FUZZ_EXAMPLES_HOSTILE=600 ZOIC_FUZZ_EXAMPLES_WILD=600 \
       ZOIC_FUZZ_EXAMPLES_UPDATE=150 ZOIC_FUZZ_EXAMPLES_SAMPLE=300 ZOIC_FUZZ_EXAMPLES_TILE=1500
timeout 3000 python -m pytest tests/test_parity_gpu.py tests/test_boundary_gpu.py tests/test_tile_gpu.py -q -m gpu -k fuzz -s 2>&1 | grep -v "^$" | tail -12
} | tee gpurun_out/fuzz_deep_r06.log
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
{
for args in "16 4096 600000 1 1 0" "4 4096 600000 1 1 0" "16 65536 20000 1 1 0" "16 256 600000 1 1 0" "16 4096 200000 0 1 0" "64 4096 100000 1 1 0 0 0 2" "128 4096 50000 1 1 0 0 0 2" "16 65536 30000 1 1 0 1 1" "4 65536 20000 1 1 4" "8 4096 300000 1 1 4"; do
  timeout -k 5 900 tools/native/tile_latency $LENS $args 2>&1 | tail -1
done
for t in tessar_f2.8 petzval_f1.25; do timeout -k 5 600 tools/native/tile_latency zoic_amd/lenses/$t.dat 16 4096 300000 1 1 0 2>&1 | tail -1; done
timeout 300 python tools/soak_mailbox.py 60 2>&1 | tail -3
} | tee gpurun_out/tile_soak_r06.txt
