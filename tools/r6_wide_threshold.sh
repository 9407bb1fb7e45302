#!/bin/bash
# tools/r6_wide_threshold.sh -- from how many samples on should a RAYTRACED tile run 64-ray (wide) batches?  default (16384) against -DZOIC_TILE_WIDE_SAMPLES=8192 / 4096 builds
cd $GRAFT_REPO_ROOT
L=zoic_amd/lenses/double_gauss_f2.0.dat
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1: %7.1f Mrays/s  p50 %8.2f  p99 %8.2f us' % (d['mrays_s'], d['p50_us'], d['p99_us']))"; }
for rep in 1 2; do for lib in default w8k w4k; do
  if [ $lib = default ]; then unset LD_PRELOAD; else export LD_PRELOAD=$PWD/tools/ubench/libzoic_$lib.so; fi
  tools/native/tile_latency $L 1 4096 2000 1 1 0 | p "$lib  1 x 4096"
  tools/native/tile_latency $L 4 4096 1000 1 1 0 | p "$lib  4 x 4096"
  tools/native/tile_latency $L 16 4096 600 1 1 0 | p "$lib 16 x 4096"
  tools/native/tile_latency $L 1 8192 1500 1 1 0 | p "$lib  1 x 8192"
  tools/native/tile_latency $L 16 8192 400 1 1 0 | p "$lib 16 x 8192"
  tools/native/tile_latency $L 16 8192 400 1 1 0 1 1 | p "$lib 16 x 8192 records+samples16"
  unset LD_PRELOAD
done; done
