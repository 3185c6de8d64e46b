#!/bin/bash
# chain-tile triangular solve v2 (fetcher wave): parity forced on the goldens, full GPU suite (header refactor), timings
mkdir -p gpurun_out/r02c
cd /root/repo
export TMPDIR=/tmp
K="lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -k "$K" > gpurun_out/r02c/forced_ct.log 2>&1; echo "forced ct tests rc=$?"; tail -3 gpurun_out/r02c/forced_ct.log
for seg in 32 16; do
 for mat in shell poisson; do
  RAMD_TRSV_CT_SEG=$seg timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02c/bench_${mat}_seg$seg.json 2> gpurun_out/r02c/bench_${mat}_seg$seg.err; echo "bench $mat seg $seg rc=$?"
 done
done
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r02c/full_gpu.log 2>&1; echo "full gpu suite rc=$?"; tail -4 gpurun_out/r02c/full_gpu.log
