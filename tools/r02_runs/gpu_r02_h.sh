#!/bin/bash
mkdir -p gpurun_out/r02h
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
for fw in 0 1; do
RAMD_TRSV_CT_FETCHER=$fw RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_ROWS=96 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02h/forced_ct_fw$fw.log 2>&1; echo "forced ct tests fetcher=$fw rc=$?"; tail -2 gpurun_out/r02h/forced_ct_fw$fw.log
done
for cfg in "0 512" "0 1000" "0 343" "1 512" "0 1728"; do
 set -- $cfg
 for mat in poisson shell; do
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_FETCHER=$1 RAMD_TRSV_CT_ROWS=$2 timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02h/bench_${mat}_fw$1_r$2.json 2> gpurun_out/r02h/bench_${mat}_fw$1_r$2.err; echo "bench $mat fetcher $1 rows $2 rc=$?"
 done
done
