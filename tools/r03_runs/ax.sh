#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03ax; mkdir -p $O
export TMPDIR=/tmp
cd $R
for i in 1 2 3; do
RAMD_ALLOC_VERBOSE=1 timeout 1200 python bench.py > $O/cg_default$i.json 2> $O/cg_default$i.err; echo "default rc=$?"
timeout 900 python bench.py --force-global --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras > $O/global$i.json 2> $O/global$i.err; echo "global1 rc=$?"
done
grep -h "place by trial" $O/cg_default1.err | cut -c1-200 | head -6
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03ax/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'place', d.get('placement_s'), (d.get('roofline') or {}).get('avg_ms'), (d.get('roofline') or {}).get('frac'), ((d.get('kernels') or {}).get('vector_updates') or {}).get('avg_ms'), 'cols', (d.get('columns_read') or {}).get('iters_per_s'), (d.get('roofline_columns_read') or {}).get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
