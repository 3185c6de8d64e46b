#!/bin/bash
mkdir -p gpurun_out/r02o
cd /root/repo
export TMPDIR=/tmp
for g in 1 2 4 8; do
  RAMD_TRSV_NOFILL=1 RAMD_TRSV_CT_GROUP=$g RAMD_TRSV_CT_LDS=65536 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02o/b_$g.json 2> gpurun_out/r02o/b_$g.err
  echo "nofill group=$g rc=$?"
  python -c "import sys,json; d=json.loads(open('gpurun_out/r02o/b_$g.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'])"
done
