# round 4: coupled aggregation AMG on the GlobalMatrix -- the aggregation and AMG tests of the distributed suite
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04ze
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "aggregation_amg or distributed_pmis" > $O/amg.log 2>&1
tail -30 $O/amg.log
