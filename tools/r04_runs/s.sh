# round 4, call 19: the 512^3 build soak 20 times with the fixed publication (both switches on = default)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04s
mkdir -p $O
cd $R
ok=0; bad=0
for rep in $(seq 1 20); do
  timeout 300 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "soak" > $O/soak_$rep.log 2>&1 && ok=$((ok+1)) || bad=$((bad+1))
done
echo "soak: $ok passed, $bad failed"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_shell.py tests/test_gpu_box_tiles_forced.py -m gpu -q -x -k "lusolve or trisolve or ilu or sgs or ic or shell or box_tiles or llsolve or lsolve or usolve" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
