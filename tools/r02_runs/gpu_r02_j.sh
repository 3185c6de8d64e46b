#!/bin/bash
mkdir -p gpurun_out/r02j
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_ROWS=96 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02j/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -2 gpurun_out/r02j/forced_ct.log
for rows in 512 1000; do
 for mat in poisson shell; do
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=$rows timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02j/bench_${mat}_r$rows.json 2> gpurun_out/r02j/bench_${mat}_r$rows.err; echo "bench $mat rows $rows rc=$?"
 done
done
bash tools/profile_r02.sh > gpurun_out/r02j/profile.log 2>&1; tail -5 gpurun_out/r02j/profile.log
python tools/prof_summary.py r02 > gpurun_out/r02j/summary.log 2>&1
mkdir -p gpurun_out/r02j/profiles && cp profiles/r02_* gpurun_out/r02j/profiles/
