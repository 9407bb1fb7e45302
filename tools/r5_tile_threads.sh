#!/bin/bash
# tools/r5_tile_threads.sh [tag] -- bucket-sized tiles from 1 / 2 / 4 / 8 / 16 render threads (what a renderer does): Mrays/s and latency per thread count
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
LENS=zoic_amd/lenses/double_gauss_f2.0.dat
for n in 4096 16384 65536; do for t in 1 2 4 8 16; do
  timeout -k 5 120 tools/native/tile_latency $LENS $t $n $((n == 65536 ? 100 : 600)) 1 1 0 2>&1 | tail -1
done; done | tee gpurun_out/tile_threads_${1:-x}.txt | python tools/tile_table.py /dev/stdin
# the same with zoic_ray records (zoic_tile_set_rows: 32 instead of 84 bytes a ray leave the GPU)
for n in 4096 65536; do for t in 1 4 16; do
  timeout -k 5 120 tools/native/tile_latency $LENS $t $n $((n == 65536 ? 100 : 600)) 1 1 0 1 2>&1 | tail -1
done; done | tee gpurun_out/tile_threads_rays_${1:-x}.txt | python tools/tile_table.py /dev/stdin
