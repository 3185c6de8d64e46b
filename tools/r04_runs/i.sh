# round 4, call 9: k_csr_w4p (look-ahead with static sets) against k_csr_w4 on the config-3 surrogate
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_forced and W4" > $O/pytest.log 2>&1
echo "pytest w4 rc=$?"; tail -2 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_shell.py -m gpu -q -x > $O/pytest_shell.log 2>&1
echo "pytest shell rc=$?"; tail -2 $O/pytest_shell.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for pf in 0 1 4; do
    RAMD_CSR_W4_PF=$pf TAG=pf$pf timeout 300 python $R/tools/spmv_shell.py 549 2>&1 | grep "^shell" >> $O/spmv.log
  done
done
cat $O/spmv.log
