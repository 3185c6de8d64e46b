#!/bin/bash
O=gpurun_out/r03ay; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "placement_entry" > $O/t.log 2>&1; echo "rc=$?"; tail -15 $O/t.log
