#!/bin/bash
# tools/r6_whatif.sh -- same-box timing of the TIMING-ONLY what-if builds (-DZOIC_EXP_WHATIF=n, kolb_pool_body.hpp: their rays are wrong on purpose):
# what would C3 gain without its retries' gathers (1) / without the first try's probe gather (2), C2 / C3 without the listed kernel (3),
# C2 without the dead list's sample re-read (4)?
cd $GRAFT_REPO_ROOT
run() { # lib config precision
  if [ "$1" = default ]; then unset ZOIC_AMD_LIB; else export ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$1.so; fi
  python bench.py --only-headline --config $2 --precision $3 --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', d['value'], d['ms_per_step'])"
}
for rep in 1 2; do
  for lib in default wi1 wi2 wi3; do run $lib C3 fast; done
  for lib in default wi1 wi2; do run $lib C3 unchecked; done
  for lib in default wi3 ${WI4:-}; do run $lib C2 fast; done
  if [ -n "$WI4" ]; then for lib in default wi4; do run $lib C2 unchecked; done; fi
done
