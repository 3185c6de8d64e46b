R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
RAMD_BUILD_VERBOSE=1 RAMD_TRSV_CT_VERBOSE=1 TAG=build timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep -E "not used|sync-free grouped plan|tag=" | cut -c1-220
TAG=delaunay timeout 600 python tools/sf_check.py delaunay 549 2 2>&1 | grep -E "tag=" | cut -c1-220
timeout 1200 python -m pytest tests/test_gpu_syncfree.py tests/test_gpu_shell.py tests/test_gpu_box_tiles_forced.py -m gpu -x -q 2>&1 | tail -3
