#!/bin/bash
# tools/r6_ab.sh "<configs>" "<precisions>" <reps> <lib>... -- same-box A/B of library builds, interleaved repetitions ("default" = zoic_amd/libzoic_amd.so)
cd $GRAFT_REPO_ROOT
CFGS=$1; PRECS=$2; REPS=$3; shift 3
for rep in $(seq 1 $REPS); do for c in $CFGS; do for p in $PRECS; do for lib in "$@"; do
  if [ "$lib" = default ]; then unset ZOIC_AMD_LIB; else export ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$lib.so; fi
  python bench.py --only-headline --config $c --precision $p --steps ${STEPS:-20} --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$rep $lib $c $p', d['value'], d['ms_per_step'])"
done; done; done; done
