#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c3_placement2.log; : > $O
rocm-smi --showmemorypartition --showcomputepartition >> $O 2>&1
for p in 1 2; do echo "== process $p" >> $O; timeout 300 python tools/r6_c3_placement2.py >> $O 2>&1; done
cat $O
