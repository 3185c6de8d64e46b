R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python -m pytest tests/test_gpu_shell.py tests/test_gpu_syncfree.py tests/test_gpu_box_tiles_forced.py tests/test_gpu_lattice.py -m gpu -x -q 2>&1 | tail -3
