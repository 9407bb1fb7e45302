#!/bin/bash
# candidates pushed apart by large spacer allocations: do the pairs reach the fast class on a box where four plain candidates are all slow?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/spacers.log; : > $O
for rep in 1 2; do
  for sp in 0 8 24 48; do
    SPACER_GB=$sp timeout 300 python tools/r6_placement_matrix.py C3 fast 4 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
