#!/bin/bash
O=gpurun_out/r03ar; mkdir -p $O
export TMPDIR=/tmp
for D in 0 1; do DOT=$D TAG=xrow timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1; done
RAMD_CSR_PAT=0 DOT=1 TAG=xrow-colsread timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1
RAMD_CSR_PAT=0 DOT=0 TAG=xrow-colsread timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1
DOT=1 TAG=xrow timeout 300 python tools/spmv_time.py 256 200 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu > $O/t1.log 2>&1; echo "kernel+solver tests rc=$?"; tail -3 $O/t1.log
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03ar/a*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'], 'cols', (d.get('columns_read') or {}).get('iters_per_s'), (d.get('roofline_columns_read') or {}).get('frac'))
PY
