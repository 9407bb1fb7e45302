cd $GRAFT_REPO_ROOT
python tools/kbench.py --modes strict,fast --tag lean64 2>&1 | grep -v amdgpu | grep -v KBENCH
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
