#!/bin/bash
mkdir -p gpurun_out/r02av
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_VERBOSE=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -s -m gpu -k "27_point" > gpurun_out/r02av/t.log 2>&1; echo "rc=$?"; grep "box-tile plan (" gpurun_out/r02av/t.log | tail -2 | cut -c1-260; tail -3 gpurun_out/r02av/t.log
