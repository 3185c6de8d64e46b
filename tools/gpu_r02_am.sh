#!/bin/bash
mkdir -p gpurun_out/r02am
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_box_tiles_forced.py -x -q -m gpu > gpurun_out/r02am/forced.log 2>&1; echo "forced test rc=$?"; tail -5 gpurun_out/r02am/forced.log
