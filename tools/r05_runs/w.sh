R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05w
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_syncfree.py tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py tests/test_gpu_lattice.py -m gpu -x -q --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -22 $O/tests.log
