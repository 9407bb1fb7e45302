cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/profile_all.sh r05 2>&1 | tail -12
python tools/roofline_table.py r05 > gpurun_out/roofline_table_r05.md 2>&1; tail -12 gpurun_out/roofline_table_r05.md
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_r05.json
