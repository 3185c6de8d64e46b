#!/bin/bash
mkdir -p gpurun_out/r02ap
cd /root/repo
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r02ap/full_gpu.log 2>&1; echo "full gpu suite rc=$?"; tail -4 gpurun_out/r02ap/full_gpu.log
python __graft_entry__.py smoke > gpurun_out/r02ap/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r02ap/smoke.log
bash tools/profile_r02.sh > gpurun_out/r02ap/profile.log 2>&1; echo "profile rc=$?"
cd /root/repo
timeout 900 python bench.py > gpurun_out/r02ap/bench_cg.json 2> gpurun_out/r02ap/bench_cg.err; echo "bench cg rc=$?"
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --cpu-grid 256 --cpu-iters 20 > gpurun_out/r02ap/bench_gmres.json 2> gpurun_out/r02ap/bench_gmres.err; echo "bench gmres rc=$?"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 > gpurun_out/r02ap/bench_shell.json 2> gpurun_out/r02ap/bench_shell.err; echo "bench shell rc=$?"
timeout 900 python bench.py --solver bicgstab --precond mcsgs --format ell --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ap/bench_c4.json 2> gpurun_out/r02ap/bench_c4.err; echo "bench c4 rc=$?"
timeout 900 python bench.py --solver mixed --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ap/bench_c5.json 2> gpurun_out/r02ap/bench_c5.err; echo "bench c5 rc=$?"
timeout 600 python bench.py --force-global --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ap/bench_global1.json 2> gpurun_out/r02ap/bench_global1.err; echo "bench global rc=$?"
for f in cg gmres shell c4 c5 global1; do python -c "import json; d=json.loads(open('gpurun_out/r02ap/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('avg_ms'), (d.get('cpu_baseline') or {}).get('value'))"; done
