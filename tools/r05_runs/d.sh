R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 1200 python -m pytest tests/test_gpu_syncfree.py -m gpu -x -q 2>&1 | tail -30
