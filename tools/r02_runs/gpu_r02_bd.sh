#!/bin/bash
mkdir -p gpurun_out/r02bd
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/r02bd/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02bd/t.log
timeout 900 python tools/stress_trsv.py 128 2000 2>&1 | tail -3
