#!/bin/bash
O=gpurun_out/r03cb; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2 3; do
for v in 0 1; do
TAG=pipe$v RAMD_CSR_PIPE=$v timeout 300 python tools/spmv_shell.py 549 2>&1 | tail -2
done; done
RAMD_CSR_PIPE=1 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shell.py -x -q -m gpu -k "spmv or apply or shell" > $O/t1.log 2>&1; echo "spmv tests (pipe forced) rc=$?"; tail -3 $O/t1.log
