cd $GRAFT_REPO_ROOT
python tools/bench_host_path.py 2>&1 | grep -E "pinned|pageable|registered"
python -m pytest tests/test_boundary_gpu.py tests/test_sanitizers.py -m gpu -q -x 2>&1 | tail -3
