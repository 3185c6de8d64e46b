#!/bin/bash
mkdir -p gpurun_out/r02aj
cd /root/repo
export TMPDIR=/tmp
for rows in 512 384 256 768 160; do
  RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_ROWS=$rows timeout 600 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02aj/b_$rows.json 2> gpurun_out/r02aj/b_$rows.err
  echo "rows=$rows rc=$?"; grep "box-tile plan (lower)" gpurun_out/r02aj/b_$rows.err | tail -1 | cut -c60-260; python -c "import sys,json; d=json.loads(open('gpurun_out/r02aj/b_$rows.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['build_s'])"
done
timeout 600 python bench.py --solver gmres --precond ilu0 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02aj/p.json 2> gpurun_out/r02aj/p.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02aj/p.json').read().strip().splitlines()[-1]); print('poisson', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['build_s'])"
