#!/bin/bash
mkdir -p gpurun_out/r02av
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu_lusolve_poisson" > gpurun_out/r02av/t.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r02av/t.log
