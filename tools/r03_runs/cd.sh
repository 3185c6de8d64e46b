#!/bin/bash
O=gpurun_out/r03cd; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
for v in 0 1; do
TAG=w4_$v RAMD_CSR_W4=$v timeout 300 python tools/spmv_shell.py 549 2>&1 | tail -2
done; done
RAMD_CSR_W4=1 RAMD_CSR_PAT=0 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shell.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "spmv or apply or shell or jacobi or fixed" > $O/t1.log 2>&1; echo "spmv tests (w4 forced, no patterns) rc=$?"; tail -3 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shell.py -x -q -m gpu -k "spmv or apply or shell" > $O/t2.log 2>&1; echo "spmv tests (default) rc=$?"; tail -3 $O/t2.log
