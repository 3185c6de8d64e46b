cd /root/repo
RAMD_TRSV_BAND=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_shell.py -x -q -m gpu -k "(ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0 or variants) and not full_size and not cpp and not fresh_process and not forced" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_shell.py -x -q -m gpu 2>&1 | tail -3
