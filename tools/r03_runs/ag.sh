#!/bin/bash
O=gpurun_out/r03ag; mkdir -p $O
export TMPDIR=/tmp
RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases_shell.py > $O/phases_shell.log 2>&1; grep -v "^alloc" $O/phases_shell.log | grep "sweep\|schedule\|plan: levels\|GMRES" | tail -9
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | grep "sweep\|schedule\|GMRES" | tail -9
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 500 > $O/phases500.log 2>&1; grep -v "^alloc" $O/phases500.log | grep "sweep\|schedule\|GMRES" | tail -9
timeout 600 python tools/build_phases.py 512 > $O/plain.log 2>&1; tail -2 $O/plain.log
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/gpu_suite.log
