#!/bin/bash
# tools/ab.sh "<label>" [bench args]: print value / kernel ms for C3,C4 fast (+ C3 strict)
L=$1; shift
for c in C3 C4 C2 C5; do python bench.py --steps 8 --warmup 2 --config $c --precision fast --no-cpu-baseline --no-parity "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['workload'][:2], d['config']['precision_mode'], d['value'], d['roofline']['kernel_ms'])"; done
python bench.py --steps 8 --warmup 2 --config C3 --precision strict --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['workload'][:2], d['config']['precision_mode'], d['value'], d['roofline']['kernel_ms'])"
