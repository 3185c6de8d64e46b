#!/bin/bash
mkdir -p gpurun_out/r02aw
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_global_full_size.py -x -q -m gpu > gpurun_out/r02aw/t.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r02aw/t.log
