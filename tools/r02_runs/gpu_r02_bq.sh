#!/bin/bash
# do power-of-two array sizes (512^3 = 2^27 rows: every vector exactly 1 GiB) cost bandwidth?  CG bench at 512 and 504, alternating
mkdir -p gpurun_out/r02bq
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2 3; do for g in 512 504; do
timeout 900 python bench.py --grid $g --steps 150 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bq/c_${g}_$rep.json 2> gpurun_out/r02bq/c_${g}_$rep.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bq/c_${g}_$rep.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; r=d['roofline']; print('grid $g', d['value'], d['ms_per_step'], 'spmv', r['avg_ms'], r['achieved'], 'vec', v['avg_ms'], v['achieved'])"
done; done
