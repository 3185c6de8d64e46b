#!/bin/bash
mkdir -p gpurun_out/r02at
cd /root/repo
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r02at/full_gpu.log 2>&1; echo "full gpu suite rc=$?"; tail -4 gpurun_out/r02at/full_gpu.log
python __graft_entry__.py smoke > gpurun_out/r02at/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r02at/smoke.log
timeout 900 python bench.py > gpurun_out/r02at/bench_cg.json 2> gpurun_out/r02at/bench_cg.err; echo "bench cg rc=$?"
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --cpu-grid 256 --cpu-iters 20 > gpurun_out/r02at/bench_gmres.json 2> gpurun_out/r02at/bench_gmres.err; echo "bench gmres rc=$?"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 > gpurun_out/r02at/bench_shell.json 2> gpurun_out/r02at/bench_shell.err; echo "bench shell rc=$?"
for f in cg gmres shell; do python -c "import json; d=json.loads(open('gpurun_out/r02at/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_ms'), (d.get('cpu_baseline') or {}).get('value'), d.get('build_s'))"; done
