#!/bin/bash
# One box: the GPU parity suite, then kbench per library (LIBS = space-separated names under tools/ubench/libzoic_<name>.so;
# "default" = zoic_amd/libzoic_amd.so), each REPS times so that box drift shows.  CONFIGS / MODES / FLIPS as kbench.
cd /root/repo
if [ -z "$NOTEST" ]; then timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5; fi
for r in $(seq 1 ${REPS:-1}); do
for v in ${LIBS:-default}; do
  if [ "$v" = default ]; then unset ZOIC_AMD_LIB; else export ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$v.so; fi
  timeout 600 python tools/kbench.py --configs ${CONFIGS:-C2,C3,C4,C5} --modes ${MODES:-fast,unchecked,strict} --steps 10 --flips ${FLIPS:-2000000} --tag $v 2>&1 | grep -E "Grays|flips"
done; done
