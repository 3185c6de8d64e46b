#!/bin/bash
mkdir -p gpurun_out/r02aw
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "parallel_manager or distribute_matrix" > gpurun_out/r02aw/t.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/r02aw/t.log
