#!/bin/bash
O=gpurun_out/r03bk; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for rep in 1 2; do for U in 0 1 3 4; do
  RAMD_MGS_UO=$U timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 $B > $O/u${U}_$rep.json 2> $O/u${U}_$rep.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bk/u*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    k=d['kernels']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'trsv', d['roofline']['avg_ms'], 'mgs', k['vector_updates']['avg_ms'], k['vector_updates']['frac'], 'spmv', k['spmv']['avg_ms'])
PY
