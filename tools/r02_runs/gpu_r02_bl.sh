#!/bin/bash
# CSR SpMV variants (register prefetch of the next LDS pass: RAMD_CSR_PF; four lanes per row: RAMD_CSR_Q4): tests, then A/B
# on the shell surrogate (spmv avg_ms of the bench line)
mkdir -p gpurun_out/r02bl
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shell.py -x -q -m gpu -k "spmv or csr or apply or shell or variants" > gpurun_out/r02bl/t1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02bl/t1.log
for q in 0 1 0 1; do
RAMD_CSR_PF=$q timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bl/s_$q.json 2> gpurun_out/r02bl/s_$q.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bl/s_$q.json').read().strip().splitlines()[-1]); k=d['kernels']['spmv']; print('pf=$q shell', d['value'], d['ms_per_step'], k['avg_ms'], k['min_ms'], k['frac'], d['final_residual'])"
done
for q in 0 1; do
RAMD_CSR_PF=$q timeout 900 python bench.py --grid 256 --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bl/c_$q.json 2> gpurun_out/r02bl/c_$q.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bl/c_$q.json').read().strip().splitlines()[-1]); k=d['roofline']; print('pf=$q cg256', d['value'], d['ms_per_step'], k['avg_ms'], k['min_ms'], k['frac'], d['final_residual'])"
done
