# round 4, call 33: the AMG goldens and the distributed suite after the coupled aggregation went in
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zf
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_solvers.py -m gpu -x -q -k "amg or AMG or multigrid" > $O/solvers_amg.log 2>&1
tail -5 $O/solvers_amg.log
timeout 2400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_global_full_size.py -m gpu -x -q > $O/dist.log 2>&1
tail -8 $O/dist.log
