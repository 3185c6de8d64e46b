#!/bin/bash
O=gpurun_out/r03cg; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py tests/test_gpu_solvers.py -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or llsolve or box or tri or ic or golden" > $O/t1.log 2>&1; echo "trsv tests rc=$?"; tail -3 $O/t1.log
for i in 1 2 3; do
for v in 0 1; do
TAG=prefill$v RAMD_TRSV_PREFILL=$v timeout 600 python tools/trsv_time.py poisson 512 2>&1 | tail -1 | cut -d'|' -f2-
done; done
for v in 0 1; do
TAG=prefill$v RAMD_TRSV_PREFILL=$v timeout 600 python tools/trsv_time.py shell 549 2>&1 | tail -1 | cut -d'|' -f2-
TAG=prefill$v RAMD_TRSV_PREFILL=$v timeout 600 python tools/trsv_time.py poisson 256 2>&1 | tail -1 | cut -d'|' -f2-
done
