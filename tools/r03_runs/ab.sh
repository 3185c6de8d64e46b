#!/bin/bash
O=gpurun_out/r03ab; mkdir -p $O
export TMPDIR=/tmp
RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases_shell.py > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | tail -34
