#!/bin/bash
O=gpurun_out/r03ch; mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
timeout 1500 python bench.py > $O/default.json 2> $O/default.err; echo "default bench rc=$? in $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03ch/default.json').read().splitlines() if l.startswith('{')][-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_columns_read',{}).get('frac'), 'placement', d.get('placement_s'))
print('cpu', d['cpu_baseline'])
for k,v in (d.get('extras') or {}).items():
    if isinstance(v, dict):
        print(k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'), (v.get('roofline') or {}).get('avg_ms'))
PY
