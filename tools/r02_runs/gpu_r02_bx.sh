#!/bin/bash
# occupancy of the row-pattern SpMV: 7 waves per SIMD (69 VGPRs) against 8 (64 VGPRs, 5-8 spilled)
mkdir -p gpurun_out/r02bx
cd /root/repo
export TMPDIR=/tmp
run() {
for rep in 1 2; do
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bx/c_$1_$rep.json 2> gpurun_out/r02bx/c_$1_$rep.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bx/c_$1_$rep.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; r=d['roofline']; print('$1 cg', d['value'], d['ms_per_step'], 'spmv', r['avg_ms'], r['min_ms'], 'vec', v['avg_ms'])"
done
}
run w7
RAMD_EXTRA_CXXFLAGS="-DRAMD_CSR_PAT_WAVES=8" python -m rocalution_amd.build --force > gpurun_out/r02bx/rebuild.log 2>&1; echo "rebuild rc=$?"
run w8
