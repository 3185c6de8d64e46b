#!/bin/bash
O=gpurun_out/r03aa; mkdir -p $O
export TMPDIR=/tmp
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | tail -22
timeout 600 python tools/build_phases.py 512 > $O/plain.log 2>&1; tail -2 $O/plain.log
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu --no-cpu-baseline > $O/shell.json 2> $O/shell.err; echo "shell rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03aa/shell.json').read().strip().splitlines()[-1])
print('shell', d['value'], d['ms_per_step'], 'build', d.get('build_s'), d['roofline']['avg_ms'], d['final_residual'])
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/gpu_suite.log
