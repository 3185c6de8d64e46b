#!/bin/bash
# row-pattern CSR SpMV: tests, then the CG / GMRES bench lines with and without it on one box
mkdir -p gpurun_out/r02bw
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "row_patterns or variants_forced or spmv or csr" > gpurun_out/r02bw/t1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02bw/t1.log
for pat in 1 0 1 0; do
RAMD_CSR_PAT=$([ $pat = 1 ] && echo -1 || echo 0) timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bw/c_$pat.json 2> gpurun_out/r02bw/c_$pat.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bw/c_$pat.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; r=d['roofline']; print('pat=$pat cg', d['value'], d['ms_per_step'], 'spmv', r['avg_ms'], r['achieved'], 'vec', v['avg_ms'], d['final_residual'])"
done
RAMD_CSR_PAT=-1 timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bw/g.json 2> gpurun_out/r02bw/g.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bw/g.json').read().strip().splitlines()[-1]); print('gmres', d['value'], d['ms_per_step'], d['kernels']['spmv']['avg_ms'], d['final_residual'])"
