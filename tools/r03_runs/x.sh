#!/bin/bash
O=gpurun_out/r03x; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/build_phases.py 512 > $O/plain.log 2>&1; tail -2 $O/plain.log
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | tail -60
