# round 4, call 39: the whole GPU suite and smoke() on the tree with the coupled aggregation AMG
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zk
mkdir -p $O
cd $R
S=$SECONDS
timeout 3300 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1
echo "suite rc=$? $((SECONDS-S)) s"; tail -6 $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
