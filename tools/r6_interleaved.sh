#!/bin/bash
# contiguous partitions (8 fronts 506 MB apart) against interleaved ones (chunk c of partition p = chunk 8c + p of the batch: one front), every pair of K buffers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/interleaved.log; : > $O
for rep in 1 2; do
for cfg in "C3 fast 4" "C4 fast 2" "C2 fast 3" "C3 strict 2" "C5 fast 1" "C1 fast 3"; do
  for lib in cont inter; do
    ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$lib.so timeout 300 python tools/r6_placement_matrix.py $cfg 2>&1 | grep -v amdgpu.ids >> $O
  done
done; done
grep "lib=" $O
