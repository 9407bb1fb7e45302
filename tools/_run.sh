cd $GRAFT_REPO_ROOT
ZOIC_BENCH_SAME_GPU=1 timeout -k 10 600 python bench.py --gpus 2 --steps 1 --warmup 1 --no-sharded --sharded-timeout 500 > gpurun_out/bench_r05_rehearsal_2ranks_same_gpu.json 2> gpurun_out/rehearsal.err; echo "rehearsal rc=$?"; tail -c 300 gpurun_out/rehearsal.err; head -c 1800 gpurun_out/bench_r05_rehearsal_2ranks_same_gpu.json
