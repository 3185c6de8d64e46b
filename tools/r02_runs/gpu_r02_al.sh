#!/bin/bash
mkdir -p gpurun_out/r02al
cd /root/repo
export TMPDIR=/tmp
for v in 0 1 0 1; do
RAMD_MGS_NT=$v timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02al/b_$v.json 2> gpurun_out/r02al/b_$v.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02al/b_$v.json').read().strip().splitlines()[-1]); print('ntw=$v', d['value'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms'], d['kernels']['vector_updates']['min_ms'], d['kernels']['vector_updates']['max_ms'])"
done
