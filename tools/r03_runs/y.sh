#!/bin/bash
O=gpurun_out/r03y; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ilu or golden or trisolve or lusolve" > $O/t1.log 2>&1; echo "ilu tests rc=$?"; tail -5 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_box_tiles_forced.py -x -q -m gpu > $O/t2.log 2>&1; echo "forced box tiles rc=$?"; tail -3 $O/t2.log
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/phases.log 2>&1; grep -v "^alloc" $O/phases.log | tail -32
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu --no-cpu-baseline > $O/gmres.json 2> $O/gmres.err; echo "gmres rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03y/gmres.json').read().strip().splitlines()[-1])
print('gmres', d['value'], d['ms_per_step'], 'build', d.get('build_s'), d['roofline']['avg_ms'], d['final_residual'])
PY
