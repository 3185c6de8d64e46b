#!/bin/bash
O=gpurun_out/r03as; mkdir -p $O
export TMPDIR=/tmp
for D in 0 1; do DOT=$D TAG=xrow timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1; done
RAMD_CSR_PAT=0 DOT=1 TAG=colsread timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1
RAMD_CSR_PAT=0 DOT=0 TAG=colsread timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03as/a*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'], 'cols', (d.get('columns_read') or {}).get('iters_per_s'), (d.get('roofline_columns_read') or {}).get('frac'))
PY
