#!/bin/bash
mkdir -p gpurun_out/r02ax
cd /root/repo
export TMPDIR=/tmp
for pc in ic sgs ilu0; do
timeout 900 python bench.py --solver cg --precond $pc --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ax/b_$pc.json 2> gpurun_out/r02ax/b_$pc.err; echo "cg+$pc rc=$?"; tail -2 gpurun_out/r02ax/b_$pc.err | cut -c1-200; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ax/b_$pc.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['final_residual'], d['build_s'])"
done
timeout 900 python bench.py --solver gmres --precond ilu0 --grid 256 --steps 60 --warmup 10 --no-reference-gpu --cpu-iters 60 > gpurun_out/r02ax/b_256.json 2> gpurun_out/r02ax/b_256.err; echo "gmres 256 rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ax/b_256.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['final_residual'], d['cpu_baseline'])"
