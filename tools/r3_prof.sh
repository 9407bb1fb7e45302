#!/bin/bash
# profile the pool variant: tools/profile.sh per config, then the summaries
cd /root/repo
export ZOIC_KOLB_VARIANT=${ZOIC_KOLB_VARIANT:-pool}
for c in ${CONFIGS:-C3 C2}; do
  bash tools/profile.sh pool_$c --config $c > /dev/null 2>&1
  n=$(python -c "from zoic_amd.workloads import ray_count; print(ray_count('$c'))")
  echo "== $c"; python tools/pmc_summary.py gpurun_out/prof_pool_$c $n pool
done
