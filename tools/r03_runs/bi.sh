#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03bi; mkdir -p $O
export TMPDIR=/tmp
cd $R
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
for i in 1 2 3; do timeout 1200 python bench.py > $O/cg_default$i.json 2> $O/cg_default$i.err; echo "default rc=$?"; done
timeout 900 python bench.py --force-global --steps 100 --warmup 10 > $O/global1.json 2> $O/global1.err; echo "global1 rc=$?"
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu > $O/gmres.json 2> $O/gmres.err; echo "gmres rc=$?"
timeout 900 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/bicgstab.json 2> $O/bicgstab.err; echo "bicgstab rc=$?"
timeout 900 python bench.py --grid 256 --steps 400 --warmup 40 --no-cpu-baseline --no-reference-gpu --no-extras > $O/cg256.json 2> $O/cg256.err; echo "cg256 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bi/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'place', d.get('placement_s'), (d.get('roofline') or {}).get('avg_ms'), (d.get('roofline') or {}).get('frac'), {k:(v['avg_ms'],v['frac']) for k,v in (d.get('kernels') or {}).items()}, 'cols', (d.get('columns_read') or {}).get('iters_per_s'), (d.get('roofline_columns_read') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; grep -v "Gloo\|amdgpu.ids" $O/gpu_suite.log | tail -4
