#!/bin/bash
mkdir -p gpurun_out/r02au
cd /root/repo
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02au/cg_$i.json 2>/dev/null; python -c "import sys,json; d=json.loads(open('gpurun_out/r02au/cg_$i.json').read().strip().splitlines()[-1]); print('cg', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'], d['kernels']['vector_updates']['avg_ms'])"; done
rocm-smi --showclocks 2>/dev/null | head -20
