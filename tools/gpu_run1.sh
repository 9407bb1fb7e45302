set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 1500 python -m pytest tests/test_boundary_gpu.py tests/test_sanitizers.py -m gpu -x -q > gpurun_out/r2a/boundary.log 2>&1; echo "boundary rc $?" >> gpurun_out/r2a/boundary.log
tail -30 gpurun_out/r2a/boundary.log
timeout 600 python tools/bench_host_path.py > gpurun_out/r2a/hostpath.log 2>&1; tail -12 gpurun_out/r2a/hostpath.log
timeout 600 python tools/flip_dump.py gpurun_out/r2a/flips > gpurun_out/r2a/flips.log 2>&1; tail -5 gpurun_out/r2a/flips.log
for c in C1 C2 C3 C4 C5; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-parity > gpurun_out/r2a/bench_$c.json 2> gpurun_out/r2a/bench_$c.err; tail -c 600 gpurun_out/r2a/bench_$c.json; done
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_boundary_gpu.py --deselect tests/test_sanitizers.py > gpurun_out/r2a/gpu_all.log 2>&1; tail -5 gpurun_out/r2a/gpu_all.log
