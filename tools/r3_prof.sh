#!/bin/bash
# profile the Kolb kernels: tools/profile.sh per config (CONFIGS), then the summaries
cd /root/repo
for c in ${CONFIGS:-C3 C2}; do
  bash tools/profile.sh ${TAG:-r3}_$c --config $c ${PRECISION:+--precision $PRECISION} > /dev/null 2>&1
  n=$(python -c "from zoic_amd.workloads import ray_count; print(ray_count('$c'))")
  echo "== $c"; python tools/pmc_summary.py gpurun_out/prof_${TAG:-r3}_$c $n pool
done
