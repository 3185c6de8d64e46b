# round 4, call 16: the whole GPU suite with the final tree
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
