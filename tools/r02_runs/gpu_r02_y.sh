#!/bin/bash
mkdir -p gpurun_out/r02y
cd /root/repo
export TMPDIR=/tmp
for cfg in "729 49152" "1000 49152" "343 40960"; do
  set -- $cfg
  RAMD_TRSV_PROF=1 RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=$1 RAMD_TRSV_CT_LDS=$2 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02y/b_$1.json 2> gpurun_out/r02y/b_$1.err
  echo "rows=$1 lds=$2 rc=$?"; grep "box-tile plan (lower)" gpurun_out/r02y/b_$1.err | tail -1; grep "trsv prof (" gpurun_out/r02y/b_$1.err | tail -2
done
