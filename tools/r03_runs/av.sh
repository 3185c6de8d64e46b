#!/bin/bash
O=gpurun_out/r03av; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6 7 8 9 10; do
  RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
grep -h "place by trial" $O/a1.err $O/a2.err $O/a3.err | cut -c1-160 | head -12
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03av/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'place', d.get('placement_s'), d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'], 'cols', (d.get('columns_read') or {}).get('iters_per_s'), (d.get('roofline_columns_read') or {}).get('frac'))
PY

