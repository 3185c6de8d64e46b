#!/bin/bash
O=gpurun_out/r03bu; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or llsolve or box or tri" > $O/t1.log 2>&1; echo "trsv tests rc=$?"; tail -3 $O/t1.log
for i in 1 2; do
for v in "RAMD_TRSV_WSLOT=0 RAMD_TRSV_PACK=0" "RAMD_TRSV_WSLOT=1 RAMD_TRSV_PACK=0" "RAMD_TRSV_WSLOT=1 RAMD_TRSV_PACK=1" "RAMD_TRSV_WSLOT=0 RAMD_TRSV_PACK=1"; do
env TAG="$v" $v timeout 600 python tools/trsv_time.py poisson 512 2>&1 | tail -1
done; done
timeout 600 python tools/trsv_time.py poisson 256 2>&1 | tail -1
timeout 600 python tools/trsv_time.py shell 549 2>&1 | tail -1
bash tools/pmc_trsv.sh $O/pmc1 RAMD_TRSV_WSLOT=1 2>&1 | tail -8
