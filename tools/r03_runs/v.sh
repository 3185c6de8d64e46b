#!/bin/bash
O=gpurun_out/r03v; mkdir -p $O
export TMPDIR=/tmp
RAMD_ALLOC_VERBOSE=1 timeout 600 python tools/spmv_placement.py 2> $O/err1.log | tee $O/out1.log; grep "class probe" $O/err1.log | head -6 | cut -c1-200
timeout 600 python tools/spmv_placement.py 2>/dev/null | tee $O/out2.log
