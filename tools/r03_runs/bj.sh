#!/bin/bash
O=gpurun_out/r03bj; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6 7 8; do
  RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bj/a*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    e=[l for l in open(f[:-5]+'.err') if 'place by trial' in l][:3]
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'place', d.get('placement_s'), d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'])
    for l in e: print('     ', l.strip()[:230])
PY
