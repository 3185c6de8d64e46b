#!/bin/bash
O=gpurun_out/r03ac; mkdir -p $O
export TMPDIR=/tmp
RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases_shell.py > $O/phases_shell.log 2>&1; grep -v "^alloc" $O/phases_shell.log | tail -16
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | grep "sweep\|build" | tail -8
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu or golden or trisolve or lusolve" > $O/t1.log 2>&1; echo "ilu tests rc=$?"; tail -2 $O/t1.log
