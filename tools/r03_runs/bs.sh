#!/bin/bash
O=gpurun_out/r03bs; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py -x -q -m gpu -k "bicg or BiCG or edge or full" > $O/t1.log 2>&1; echo "bicgstab tests rc=$?"; tail -3 $O/t1.log
for i in 1 2 3 4; do
RAMD_ALLOC_VERBOSE=1 timeout 900 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras > $O/b$i.json 2> $O/b$i.err
grep -h "place by trial" $O/b$i.err | head -1 | cut -c1-200
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bs/b*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'place', d.get('placement_s'), d['roofline']['avg_ms'], d['roofline']['frac'])
PY
