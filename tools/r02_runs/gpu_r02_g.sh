#!/bin/bash
mkdir -p gpurun_out/r02g
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02g/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -3 gpurun_out/r02g/forced_ct.log
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_ROWS=48 RAMD_TRSV_CT_GROUP=3 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -k "$K" > gpurun_out/r02g/forced_ct_small_tiles.log 2>&1; echo "forced ct (48-row tiles, groups of 3) rc=$?"; tail -3 gpurun_out/r02g/forced_ct_small_tiles.log
timeout 900 python -m pytest tests/test_gpu_shell.py tests/test_gpu_edge_cases.py -x -q > gpurun_out/r02g/shell.log 2>&1; echo "shell+edge rc=$?"; tail -3 gpurun_out/r02g/shell.log
for cfg in "4 512" "1 512" "8 512" "4 216" "4 1728"; do
 set -- $cfg
 for mat in poisson shell; do
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GROUP=$1 RAMD_TRSV_CT_ROWS=$2 timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02g/bench_${mat}_g$1_r$2.json 2> gpurun_out/r02g/bench_${mat}_g$1_r$2.err; echo "bench $mat group $1 rows $2 rc=$?"
 done
done
