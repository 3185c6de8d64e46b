#!/bin/bash
mkdir -p gpurun_out/r02ab
cd /root/repo
export TMPDIR=/tmp
for cfg in "729 49152" "640 49152" "400 40960"; do
  set -- $cfg
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=$1 RAMD_TRSV_CT_LDS=$2 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ab/b_$1.json 2> gpurun_out/r02ab/b_$1.err
  echo "rows=$1 lds=$2 rc=$?"; grep "box-tile plan (lower)" gpurun_out/r02ab/b_$1.err | tail -1 | cut -c1-200; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ab/b_$1.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'])"
done
