#!/bin/bash
# tools/r5_stop.sh <tag> [lib] -- the stop's conjugate-form root (fast_optics.hpp fast_hit): raw FAST/STRICT disagreements (flip_dump unchecked,
# analysed on the CPU afterwards with tools/flip_analysis.py), the FAST parity tests, kernel times + bench lines per config
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-x}
[ -n "$2" ] && export ZOIC_AMD_LIB=$2
echo "== lib ${ZOIC_AMD_LIB:-default}"
timeout 900 python tools/flip_dump.py gpurun_out/flips_$TAG 16777216 unchecked 2>&1 | tail -5
timeout 900 python tools/flip_dump.py gpurun_out/flips_safe_$TAG 16777216 2>&1 | tail -5
bash tools/r4_quick.sh C2 C3 C4 C5 2>&1 | tail -20
