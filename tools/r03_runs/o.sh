#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scalar_programs" > $O/k.log 2>&1; echo "scalar rc=$?"; tail -3 $O/k.log
timeout 1800 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -m gpu > $O/s.log 2>&1; echo "solvers rc=$?"; tail -6 $O/s.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/gpu_suite.log
