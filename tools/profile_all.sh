#!/bin/bash
# tools/profile_all.sh <round>: rocprofv3 kernel stats + PMC passes for every benchmarked (config, mode), then the judged
# summaries copied into profiles/ (run on the GPU box; gpurun_out/profiles_<round>.tar carries them back).
R=${1:-r03}
cd $GRAFT_REPO_ROOT
for CP in "C3 fast" "C3 unchecked" "C3 strict" "C2 fast" "C4 fast" "C5 fast" "C1 fast"; do
  set -- $CP
  # sub-millisecond frames: enough warm-up launches to leave the slow first ~25 ms after an idle gap (bench.py's per-config steps)
  case $1 in C2) export ZOIC_PROFILE_STEPS=200 ZOIC_PROFILE_WARMUP=60;; C1) export ZOIC_PROFILE_STEPS=400 ZOIC_PROFILE_WARMUP=300;; *) unset ZOIC_PROFILE_STEPS ZOIC_PROFILE_WARMUP;; esac
  bash tools/profile.sh ${R}_$2_$1 --config $1 --precision $2 > /dev/null 2>&1
  n=$(python -c "from zoic_amd.workloads import ray_count; print(ray_count('$1'))")
  python tools/collect_profiles.py ${R}_$2_$1 $n $1_$2 > gpurun_out/prof_${R}_$2_$1/collect.log 2>&1
  echo "profiled $1 $2: $(cat gpurun_out/prof_${R}_$2_$1/errors.txt 2>/dev/null) $(grep -E 'bench line|timed dispatches' gpurun_out/prof_${R}_$2_$1/collect.log | tr '\n' ';')"
done
tar cf gpurun_out/profiles_$R.tar profiles/${R}_* profiles/pmc_traffic.json
