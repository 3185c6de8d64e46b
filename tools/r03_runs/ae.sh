#!/bin/bash
O=gpurun_out/r03ae; mkdir -p $O
export TMPDIR=/tmp
for C in 64 256 1024 4096; do
echo "poll cap $C"
RAMD_SWEEP_POLLCAP=$C RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases_shell.py 549 > $O/p_$C.log 2>&1; grep "sweep  \|plan: levels\|coordinates\|GMRES" $O/p_$C.log | tail -4
RAMD_SWEEP_POLLCAP=$C RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases.py 512 > $O/q_$C.log 2>&1; grep "sweep  \|coordinates\|ilu0\|GMRES" $O/q_$C.log | tail -5
done
