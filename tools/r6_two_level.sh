#!/bin/bash
# tools/r6_two_level.sh -- same box, interleaved: the two-level retry search (-DZOIC_TWO_LEVEL_SEARCH=n, kolb_pool_body.hpp) against the shipped search
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in default tl2 tl3; do
  if [ "$lib" = default ]; then unset ZOIC_AMD_LIB; else export ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$lib.so; fi
  for c in C2 C5; do for p in fast unchecked strict; do
    if [ $c = C5 ] && [ $p = strict ]; then continue; fi
    python bench.py --only-headline --config $c --precision $p --steps ${STEPS:-20} --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $c $p', d['value'], d['ms_per_step'])"
  done; done
done; done
