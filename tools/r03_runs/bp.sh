#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03bp; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; grep -v "Gloo\|amdgpu.ids" $O/gpu_suite.log | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2 3; do timeout 1200 python bench.py > $O/cg_default$i.json 2> $O/cg_default$i.err; echo "default rc=$?"; done
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu > $O/shell.json 2> $O/shell.err; echo "shell rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bp/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'build', d.get('build_s'), 'place', d.get('placement_s'), (d.get('roofline') or {}).get('avg_ms'), (d.get('roofline') or {}).get('frac'), {k:(v['avg_ms'],v['frac']) for k,v in (d.get('kernels') or {}).items()}, 'cols', (d.get('columns_read') or {}).get('iters_per_s'), (d.get('roofline_columns_read') or {}).get('frac'))
        if d.get('extras'): print('   extras', {k:(v.get('iters_per_s') or v.get('value'), v.get('ms_per_step'), v.get('build_s')) for k,v in d['extras'].items()})
    except Exception as e: print(f, 'ERR', e)
PY
