R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05x
mkdir -p $O
cd $R
export RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 RAMD_TRSV_BAND=0
SEL="((ilu or lusolve or trisolve or preconditioner_apply) and not full_size and not cpp and not fresh_process and not gmres30 and not solvers_vs_golden) or variants_of_the_class or bit_exact_vs_oracle"
( time timeout 1200 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_shell.py -k "$SEL" --durations=15 ) > $O/inner.log 2>&1
tail -30 $O/inner.log
