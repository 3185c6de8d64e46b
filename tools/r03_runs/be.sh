#!/bin/bash
O=gpurun_out/r03be; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2; do
  RAMD_CSR_PAT2=1 RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/v$i.json 2> $O/v$i.err
  grep -h "place by trial" $O/v$i.err | cut -c1-170 | head -4
done
# is it the alternation? time each kernel kind alone and interleaved, same vectors (tools/cg_interplay.py)
for P in 0 1; do RAMD_CSR_PAT2=$P timeout 600 python tools/cg_interplay.py 2>&1 | tail -6; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03be/v*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'])
PY
