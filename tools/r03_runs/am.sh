#!/bin/bash
O=gpurun_out/r03am; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for T in 6 10; do
for i in 1 2 3 4 5 6; do
  t0=$(date +%s.%N)
  RAMD_PLACE_TRIES=$T RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/t${T}_$i.json 2> $O/t${T}_$i.err
  echo "$(echo "$(date +%s.%N) - $t0" | bc) s wall" > $O/t${T}_$i.time
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03am/t*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        e=open(f[:-5]+'.err').read()
        moved=e.count('-> moved')
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms'], 'moved', moved, open(f[:-5]+'.time').read().strip())
    except Exception as e: print(f, 'ERR', e)
PY
