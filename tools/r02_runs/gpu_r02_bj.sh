#!/bin/bash
# A/B of the consumer-class position order on one box (512^3 GMRES+ILU0, trsv average per triangle)
mkdir -p gpurun_out/r02bj
cd /root/repo
export TMPDIR=/tmp
for rep in 1 2; do for c in 0 1; do
RAMD_TRSV_CT_CLASS=$c timeout 900 python bench.py --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bj/b_$c$rep.json 2> gpurun_out/r02bj/b_$c$rep.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bj/b_$c$rep.json').read().strip().splitlines()[-1]); print('class=$c', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'], d['build_s'])"
done; done
