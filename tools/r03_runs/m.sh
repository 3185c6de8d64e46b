#!/bin/bash
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/r03_runs/m.py 2>&1 | grep -v "^$" | grep "halo all-gather\|bad rows\|Error" | head -20
RAMD_ALLOC_VERBOSE=1 timeout 300 python tools/placement_probe.py 14 2>&1 | grep -v "^addresses\|^offsets" | tee $O/probe1.log | cut -c1-600
