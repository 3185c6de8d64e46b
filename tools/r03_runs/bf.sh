#!/bin/bash
O=gpurun_out/r03bf; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6 7 8 9 10; do
  RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bf/a*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    e=[l for l in open(f[:-5]+'.err') if 'place by trial' in l][:2]
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'], 'cols', (d.get('columns_read') or {}).get('iters_per_s'))
    for l in e: print('     ', l.strip()[:150])
PY
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_full_size.py -x -q -m gpu > $O/t.log 2>&1; echo "tests rc=$?"; tail -3 $O/t.log
