#!/bin/bash
mkdir -p gpurun_out/r02ac
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02ac/forced_ct.log 2>&1; echo "forced ct rc=$?"; tail -3 gpurun_out/r02ac/forced_ct.log
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r02ac/full_gpu.log 2>&1; echo "full gpu suite rc=$?"; tail -4 gpurun_out/r02ac/full_gpu.log
