#!/bin/bash
O=gpurun_out/r03cq; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu_lusolve_poisson_vs_oracle" > $O/t1.log 2>&1; echo "lusolve oracle rc=$?"; tail -3 $O/t1.log
RAMD_TRSV_WSLOT=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu_lusolve_poisson_vs_oracle" > $O/t2.log 2>&1; echo "lusolve oracle (wslot) rc=$?"; tail -3 $O/t2.log
timeout 600 python tools/stress_trsv.py 100 600 2>&1 | tail -3
timeout 600 python tools/stress_trsv.py 77 600 2>&1 | tail -3
