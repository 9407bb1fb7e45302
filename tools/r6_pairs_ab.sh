#!/bin/bash
# does a build change C3's dependence on the pair of allocations?  the same 4 x 4 pairs per library, interleaved, one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pairs_ab.log; : > $O
for rep in 1 2; do
  for lib in "$@"; do
    ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$lib.so timeout 300 python tools/r6_placement_matrix.py C3 fast 4 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
