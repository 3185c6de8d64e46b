#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
timeout 3400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -12 $O/gpu_suite.log
