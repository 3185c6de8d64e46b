#!/bin/bash
O=gpurun_out/r03cp; mkdir -p $O
export TMPDIR=/tmp
for i in 1 2; do
S=$(date +%s)
timeout 1500 python bench.py > $O/default$i.json 2> $O/default$i.err; echo "default bench rc=$? in $(( $(date +%s) - S )) s"
python - $O/default$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_columns_read',{}).get('frac'), 'placement', d.get('placement_s'))
for k,v in (d.get('extras') or {}).items():
    if isinstance(v, dict):
        print(' ', k, v.get('iters_per_s'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('avg_ms'))
PY
done
