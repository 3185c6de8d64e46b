#!/bin/bash
O=gpurun_out/r03af; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu or golden or trisolve or lusolve" > $O/t1.log 2>&1; echo "ilu tests rc=$?"; tail -2 $O/t1.log
RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases_shell.py > $O/phases_shell.log 2>&1; grep -v "^alloc" $O/phases_shell.log | grep "sweep\|schedule\|plan: levels\|GMRES" | tail -9
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | grep "sweep\|schedule\|GMRES" | tail -9
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 500 > $O/phases500.log 2>&1; grep -v "^alloc" $O/phases500.log | grep "sweep\|schedule\|GMRES" | tail -9
timeout 900 python -m pytest tests/test_gpu_box_tiles_forced.py -x -q -m gpu > $O/t2.log 2>&1; echo "forced box tiles rc=$?"; tail -3 $O/t2.log
