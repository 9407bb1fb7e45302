cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/kbench.py --modes unchecked,fast --tag guard 2>&1 | grep -v amdgpu.ids | grep -v KBENCH
