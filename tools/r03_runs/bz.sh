#!/bin/bash
O=gpurun_out/r03bz; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or llsolve or box or tri" > $O/t1.log 2>&1; echo "trsv tests rc=$?"; tail -3 $O/t1.log
for i in 1 2 3 4; do
timeout 600 python tools/trsv_time.py poisson 512 2>&1 | tail -1 | cut -d'|' -f2-
done
timeout 600 python tools/trsv_time.py poisson 256 2>&1 | tail -1 | cut -d'|' -f2-
timeout 600 python tools/trsv_time.py poisson 500 2>&1 | tail -1 | cut -d'|' -f2-
timeout 600 python tools/trsv_time.py shell 549 2>&1 | tail -1 | cut -d'|' -f2-
