#!/bin/bash
# transposed phase-A records on the kernels that do NOT ship them (C2's two-level kernel, C4, C5): per-lane stores (cont) against the transpose everywhere (tr), every pair of K buffers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/tr_configs.log; : > $O
for rep in 1 2; do
for cfg in "C2 fast 3" "C4 fast 2" "C5 fast 1" "C3 strict 2"; do
  for lib in cont tr; do
    ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$lib.so timeout 300 python tools/r6_placement_matrix.py $cfg 2>&1 | grep "lib=" >> $O
  done
done; done
cat $O
