#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03bb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "^[[:space:]]*Name[[:space:]]*:[[:space:]]*[A-Za-z0-9_]*" $O/counters.txt | awk '{print $NF}' | sort -u > $O/names.txt
wc -l $O/names.txt
grep -c . $O/counters.txt
grep -i "^SQ_WAIT\|^SQ_ACTIVE\|^SQ_INSTS\|^SQ_BUSY\|^SQ_WAVE\|^TA_\|^TCP_\|^TD_\|^TCC_HIT\|^TCC_MISS\|^TCC_REQ\|^TCC_EA_RD\|^SQ_LDS\|^SQ_INST_CYCLES\|^SQ_IFETCH\|^SQ_LEVEL" $O/names.txt | tr '\n' ' '
