#!/bin/bash
# tools/r5_tile_soak.sh -- the tile server under sustained load (a hand-off that loses a batch would show as a 20 s stall and an error):
# millions of bucket-sized tiles from 16 / 4 render threads, the largest tiles, then the per-sample mailbox under churn
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
{
for args in "16 4096 1000000 1 1 0" "4 4096 1000000 1 1 0" "16 65536 30000 1 1 0" "16 256 1000000 1 1 0" "16 4096 300000 0 1 0"; do
  timeout -k 5 900 tools/native/tile_latency $LENS $args 2>&1 | tail -2
done
for t in tessar_f2.8 petzval_f1.25; do timeout -k 5 600 tools/native/tile_latency zoic_amd/lenses/$t.dat 16 4096 500000 1 1 0 2>&1 | tail -2; done
timeout 300 python tools/soak_mailbox.py 60 2>&1 | tail -3
} | tee gpurun_out/tile_soak_r05.txt
# ... and answered in zoic_ray records (zoic_tile_set_rows)
for args in "16 65536 40000 1 1 0 1" "16 4096 1000000 1 1 0 1"; do timeout -k 5 900 tools/native/tile_latency zoic_amd/lenses/double_gauss_f2.0.dat $args 2>&1 | tail -1; done | tee -a gpurun_out/tile_soak_r05.txt
