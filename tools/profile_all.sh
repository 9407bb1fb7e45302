#!/bin/bash
# tools/profile_all.sh <round>: rocprofv3 kernel stats + PMC passes for every benchmarked (config, mode); run on the GPU box.
R=${1:-r02}
cd $GRAFT_REPO_ROOT
for CP in "C3 fast" "C3 unchecked" "C3 strict" "C2 fast" "C4 fast" "C5 fast" "C1 fast"; do
  set -- $CP
  bash tools/profile.sh ${R}_$2_$1 --config $1 --precision $2 > /dev/null 2>&1
  echo "profiled $1 $2: $(cat gpurun_out/prof_${R}_$2_$1/errors.txt 2>/dev/null)"
done
