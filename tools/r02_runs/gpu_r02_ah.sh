#!/bin/bash
mkdir -p gpurun_out/r02ah
cd /root/repo
export TMPDIR=/tmp
for w in 0 1; do
RAMD_ILU0_WAVE=$w timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ah/shell_w$w.json 2> gpurun_out/r02ah/shell_w$w.err; echo "shell wave=$w rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ah/shell_w$w.json').read().strip().splitlines()[-1]); print(d['value'], d['final_residual'], 'build_s', d['build_s'])"
RAMD_ILU0_WAVE=$w timeout 900 python bench.py --solver gmres --precond ilu0 --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ah/p_w$w.json 2> gpurun_out/r02ah/p_w$w.err; echo "poisson wave=$w rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ah/p_w$w.json').read().strip().splitlines()[-1]); print(d['value'], d['final_residual'], 'build_s', d['build_s'])"
done
RAMD_ILU0_WAVE=1 timeout 900 python -m pytest tests -x -q -m gpu -k "ilu0 or ilu_ or shell" > gpurun_out/r02ah/ilu_tests_wave.log 2>&1; echo "ilu tests (wave) rc=$?"; tail -3 gpurun_out/r02ah/ilu_tests_wave.log
for i in 1 2 3; do timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ah/cg_$i.json 2>/dev/null; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ah/cg_$i.json').read().strip().splitlines()[-1]); print('cg', d['value'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms'])"; done
