#!/bin/bash
O=gpurun_out/r03z; mkdir -p $O
export TMPDIR=/tmp
for B in 128 256 512 1024; do
RAMD_ILU0_BLOCK=$B RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases$B.log 2>&1; echo "block $B"; grep "ilu0\|GMRES" $O/phases$B.log | tail -4
done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu or golden or trisolve or lusolve" > $O/t1.log 2>&1; echo "ilu tests rc=$?"; tail -2 $O/t1.log
