#!/bin/bash
# box-tile triangular solve: parity forced on the goldens, then timings (and the other round-2 kernel changes)
mkdir -p gpurun_out/r02d
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02d/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -3 gpurun_out/r02d/forced_ct.log
timeout 900 python -m pytest tests/test_gpu_shell.py -x -q > gpurun_out/r02d/shell.log 2>&1; echo "shell rc=$?"; tail -3 gpurun_out/r02d/shell.log
for rows in 512 256; do
 for mat in shell poisson; do
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=$rows timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02d/bench_${mat}_rows$rows.json 2> gpurun_out/r02d/bench_${mat}_rows$rows.err; echo "bench $mat rows $rows rc=$?"
 done
done
timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu > gpurun_out/r02d/bench_cg.json 2> gpurun_out/r02d/bench_cg.err; echo "bench cg rc=$?"
