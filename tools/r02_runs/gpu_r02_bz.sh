#!/bin/bash
# row patterns for the ELL / HYB block: tests (all SpMV / format tests with the analysis forced), then the ELL and HYB lines
mkdir -p gpurun_out/r02bz
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "row_patterns or variants_forced or spmv or csr or ell or hyb or format or convert" > gpurun_out/r02bz/t1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02bz/t1.log
for fmt in ell hyb; do for pat in -1 0; do
RAMD_CSR_PAT=$pat timeout 900 python bench.py --format $fmt --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bz/c_${fmt}_$pat.json 2> gpurun_out/r02bz/c_${fmt}_$pat.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bz/c_${fmt}_$pat.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$fmt pat=$pat cg', d['value'], d['ms_per_step'], 'spmv', r['avg_ms'], r['achieved'], d['final_residual'])"
done; done
