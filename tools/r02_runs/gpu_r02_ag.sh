#!/bin/bash
mkdir -p gpurun_out/r02ag
cd /root/repo
export TMPDIR=/tmp
bash tools/profile_r02.sh > gpurun_out/r02ag/profile.log 2>&1; echo "profile rc=$?"; tail -3 gpurun_out/r02ag/profile.log
cd /root/repo
python tools/prof_summary.py r02 > gpurun_out/r02ag/summary.log 2>&1; echo "summary rc=$?"
cp profiles/r02_* gpurun_out/r02ag/ 2>/dev/null
timeout 900 python bench.py > gpurun_out/r02ag/bench_cg.json 2> gpurun_out/r02ag/bench_cg.err; echo "bench cg rc=$?"
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --cpu-grid 256 --cpu-iters 20 > gpurun_out/r02ag/bench_gmres.json 2> gpurun_out/r02ag/bench_gmres.err; echo "bench gmres rc=$?"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 > gpurun_out/r02ag/bench_shell.json 2> gpurun_out/r02ag/bench_shell.err; echo "bench shell rc=$?"
timeout 900 python bench.py --solver bicgstab --precond mcsgs --format ell --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ag/bench_c4.json 2> gpurun_out/r02ag/bench_c4.err; echo "bench c4 rc=$?"
timeout 600 python bench.py --force-global --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ag/bench_global1.json 2> gpurun_out/r02ag/bench_global1.err; echo "bench global rc=$?"
for f in cg gmres shell c4 global1; do python -c "import json; d=json.loads(open('gpurun_out/r02ag/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'], (d.get('cpu_baseline') or {}).get('value'))"; done
