#!/bin/bash
mkdir -p gpurun_out/r02f
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02f/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -3 gpurun_out/r02f/forced_ct.log
for rows in 512 216 1000; do
 for mat in poisson shell; do
  RAMD_TRSV_CT_LDS=65000 RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=$rows timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02f/bench_${mat}_r$rows.json 2> gpurun_out/r02f/bench_${mat}_r$rows.err; echo "bench $mat rows $rows rc=$?"
 done
done
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix poisson --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02f/bench_poisson_default.json 2> gpurun_out/r02f/bench_poisson_default.err
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02f/bench_shell_default.json 2> gpurun_out/r02f/bench_shell_default.err
