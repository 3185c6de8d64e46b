#!/bin/bash
O=gpurun_out/r03ap; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "color or mc or Multi or sgs or golden" > $O/t1.log 2>&1; echo "colouring tests rc=$?"; tail -3 $O/t1.log
RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_time.py 512 > $O/build_time.log 2>&1; grep -v "^alloc" $O/build_time.log | tail -40
