#!/bin/bash
# tools/r6_listed_grid.sh -- same box: the listed kernel's grid (ZOIC_LISTED_GRID=n workgroups instead of the GUARD kernel's 2048) on the configurations
# whose work list is short (C2, C3, C5: every ray of the list is one chain of tries; the kernel's 46-50 us are launch + that chain).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for g in 0 128 256 512 1024; do
  if [ $g = 0 ]; then unset ZOIC_LISTED_GRID; else export ZOIC_LISTED_GRID=$g; fi
  for c in C2 C3; do
    python bench.py --only-headline --config $c --precision fast --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g $c fast', d['value'], d['ms_per_step'])"
  done
done; done
