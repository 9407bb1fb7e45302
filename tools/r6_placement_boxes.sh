#!/bin/bash
# what the placement probe finds on this box: the headline three times (fresh processes), and the same with adjacent candidates (no spacer) for comparison
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/placement_box.log; : > $O
for i in 1 2 3; do
  timeout 200 python bench.py --only-headline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['placement']; print('headline', d['value'], 'first', p.get('first_pair_mrays_s'), 'chosen', p.get('chosen_pair_mrays_s'), 'held_gb', p.get('held_gb'), p.get('rates_mrays_s'), p.get('seconds'))" >> $O
done
timeout 300 python -m pytest tests/test_placement_gpu.py -x -q 2>&1 | tail -2 >> $O
cat $O
