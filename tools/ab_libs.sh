#!/bin/bash
# tools/ab_libs.sh <config> <precision> lib1 lib2 ... : bench each tools/ubench/libzoic_<lib>.so (built by tools/build_variant.sh)
CFG=$1; PREC=$2; shift; shift
for L in "$@"; do
  ZOIC_AMD_LIB=$PWD/tools/ubench/libzoic_$L.so python bench.py --steps 8 --warmup 2 --config $CFG --precision $PREC --no-cpu-baseline --no-parity 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s' % '$L', '$CFG', '$PREC', d['value'], 'Grays/s  kernel_ms', d['roofline']['kernel_ms'])"
done
