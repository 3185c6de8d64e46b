#!/bin/bash
# box-tile triangular solve with grouped tickets / packed descriptors / adaptive boxes; fused vector kernels without the release fence
mkdir -p gpurun_out/r02e
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02e/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -3 gpurun_out/r02e/forced_ct.log
timeout 900 python -m pytest tests/test_gpu_shell.py -x -q > gpurun_out/r02e/shell.log 2>&1; echo "shell rc=$?"; tail -3 gpurun_out/r02e/shell.log
for grp in 4 1 2 8; do
 for mat in poisson shell; do
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GROUP=$grp timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02e/bench_${mat}_g$grp.json 2> gpurun_out/r02e/bench_${mat}_g$grp.err; echo "bench $mat group $grp rc=$?"
 done
done
timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu > gpurun_out/r02e/bench_cg.json 2> gpurun_out/r02e/bench_cg.err; echo "bench cg rc=$?"
timeout 1500 python -m pytest tests/test_gpu_global_full_size.py -x -q > gpurun_out/r02e/global_full.log 2>&1; echo "global full-size rc=$?"; tail -5 gpurun_out/r02e/global_full.log
