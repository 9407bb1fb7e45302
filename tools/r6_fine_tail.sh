#!/bin/bash
# tools/r6_fine_tail.sh -- same box: the fine tail of a launch (work_cursor.hpp: the last eighths of every cursor partition handed out in small chunks
# from a second cursor, to waves that found every coarse partition used up).  base = the library before the change; ZOIC_FINE_TAIL=eighths,shift.
cd $GRAFT_REPO_ROOT
run() { # tag lib env config precision
  if [ "$2" = default ]; then unset ZOIC_AMD_LIB; else export ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$2.so; fi
  if [ "$3" = off ]; then unset ZOIC_FINE_TAIL; else export ZOIC_FINE_TAIL=$3; fi
  python bench.py --only-headline --config $4 --precision $5 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $4 $5', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  for c in ${CONFIGS:-C2 C3 C4}; do
    run base base off $c fast
    run new_off default off $c fast
    for ft in ${TAILS:-1,2 2,2 1,1 2,3}; do run tail_$ft default $ft $c fast; done
  done
  run base base off C2 unchecked; run tail_1,2 default 1,2 C2 unchecked; run tail_2,2 default 2,2 C2 unchecked
  run base base off C2 strict; run tail_1,2 default 1,2 C2 strict
done
