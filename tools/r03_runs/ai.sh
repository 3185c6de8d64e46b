#!/bin/bash
O=gpurun_out/r03ai; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu -k "aggregation_amg" > $O/t1.log 2>&1; echo "amg tests rc=$?"; tail -40 $O/t1.log
