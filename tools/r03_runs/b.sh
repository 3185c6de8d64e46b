#!/bin/bash
# r03b: tile shape of the grouped triangular solve (one level per step), row budget, prefetch depth
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_DEDUP=1 RAMD_TRSV_CT_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_shell.py -x -q -s -m gpu -k "not full" > $O/shell_forced.log 2>&1; echo "shell forced rc=$?"; tail -2 $O/shell_forced.log
run() { TAG="$1" timeout 300 python tools/trsv_time.py shell 549 2> $O/err_$2.log | tail -1; grep "box-tile plan (" $O/err_$2.log | sed 's/.*extents/extents/'; }
RAMD_TRSV_CT_VERBOSE=1 run "default" 0
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GSHAPE=0 run "gshape0" 1
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=256 run "rows256" 2
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=1024 RAMD_TRSV_CT_LDS=65536 run "rows1024" 3
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=128 run "rows128" 4
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GROUPS=0 run "nogroups" 5
