#!/bin/bash
O=gpurun_out/r03aj; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; grep -v "Gloo\|amdgpu.ids" $O/gpu_suite.log | tail -12
