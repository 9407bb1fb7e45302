#!/bin/bash
# tools/r5_tile_timing.sh [tag] -- where a tile batch's time goes: the -DZOIC_TILE_TIMING build (tools/ubench/timing/libzoic_amd.so, built by
#   python -c "from zoic_amd import build as B; B.build(force=True, extra_flags=['-DZOIC_TILE_TIMING'], out='tools/ubench/timing/libzoic_amd.so', objdir='tools/ubench/obj_timing')")
# prints its region histograms when the camera is destroyed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out   # (mkdir -p tools/ubench/timing before building the library there)
TAG=${1:-x}
export LD_LIBRARY_PATH=$PWD/tools/ubench/timing:$LD_LIBRARY_PATH
for L in double_gauss_f2.0 tessar_f2.8; do
for prec in 1 2 0; do
  echo "== $L 300 tiles of 4096 samples, precision $prec"
  timeout -k 5 120 tools/native/tile_latency zoic_amd/lenses/$L.dat 1 4096 300 $prec 1 0 2>&1 | tail -6
done; done | tee gpurun_out/tile_timing_$TAG.txt
