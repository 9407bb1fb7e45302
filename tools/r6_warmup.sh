#!/bin/bash
# how much of a short frame's rate is the clock ramp?  the same config at bench.py's steps / warm-up and at longer ones, one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/warmup.log; : > $O
for rep in 1 2; do
for cfg in "C2 20 3" "C2 200 50" "C2 1000 200" "C1 20 3" "C1 1000 200" "C3 20 3" "C3 20 10" "C3 100 20" "C4 10 2" "C4 40 10"; do
  set -- $cfg
  v=$(timeout 200 python bench.py --only-headline --config $1 --steps $2 --warmup $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$1 steps $2 warmup $3: $v" >> $O
done; done
cat $O
