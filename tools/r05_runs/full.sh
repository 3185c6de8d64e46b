R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05full
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -25 $O/suite.log
