#!/bin/bash
# chain-tile triangular solve: parity (forced on the small goldens), then timing A/B on both matrices
mkdir -p gpurun_out/r02b
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02b/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -3 gpurun_out/r02b/forced_ct.log
timeout 900 python -m pytest tests/test_gpu_shell.py tests/test_gpu_full_size.py -x -q > gpurun_out/r02b/shell_full.log 2>&1; echo "shell+full rc=$?"; tail -3 gpurun_out/r02b/shell_full.log
for seg in 32 16; do
 for mat in shell poisson; do
  RAMD_TRSV_CT_SEG=$seg timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02b/bench_${mat}_seg$seg.json 2> gpurun_out/r02b/bench_${mat}_seg$seg.err; echo "bench $mat seg $seg rc=$?"
 done
done
