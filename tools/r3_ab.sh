#!/bin/bash
# A/B of the Kolb kernel variants on one box: parity of the pool variant, then kbench for both
cd /root/repo
export ZOIC_KOLB_VARIANT=pool
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu 2>&1 | tail -15
for v in refill pool; do
  ZOIC_KOLB_VARIANT=$v timeout 600 python tools/kbench.py --configs C2,C3,C4,C5 --modes fast,unchecked,strict --steps 10 --flips 2000000 --tag $v 2>&1 | grep -v "^$"
done
