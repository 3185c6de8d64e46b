#!/bin/bash
O=gpurun_out/r03ad; mkdir -p $O
export TMPDIR=/tmp
for B in 64 256 1024; do for NX in 275 549; do
echo "block $B nx $NX"
RAMD_SWEEP_BLOCK=$B RAMD_BUILD_VERBOSE=1 timeout 900 python tools/build_phases_shell.py $NX > $O/p_${B}_$NX.log 2>&1; grep "sweep  \|plan: levels\|coordinates\|GMRES" $O/p_${B}_$NX.log | tail -6
done; done
