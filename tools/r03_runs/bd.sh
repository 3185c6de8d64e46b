#!/bin/bash
O=gpurun_out/r03bd; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "variants_forced and PAT2" > $O/t1.log 2>&1; echo "pat2 variant tests rc=$?"; tail -2 $O/t1.log
for rep in 1 2 3 4; do for P in 0 1; do for D in 0 1; do
 RAMD_CSR_PAT2=$P DOT=$D TAG=pat2=$P timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1 | sed 's/ (min.*algorithmic = / /; s/| norm.*| /| /'
done; done; done
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4; do for P in 0 1; do
  RAMD_CSR_PAT2=$P timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/p${P}_$i.json 2> $O/p${P}_$i.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03bd/p*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d['kernels']['vector_updates']['avg_ms'])
PY
