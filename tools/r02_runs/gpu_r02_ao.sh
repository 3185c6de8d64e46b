#!/bin/bash
tag=$1
mkdir -p gpurun_out/r02ao
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02ao/kt_$tag -o b -- python $R/bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu --no-extras > $R/gpurun_out/r02ao/b_$tag.json 2> $R/gpurun_out/r02ao/b_$tag.err
cd $R; python - <<PY
import sqlite3
c=sqlite3.connect('gpurun_out/r02ao/kt_$tag/b_results.db').cursor()
for r in c.execute("select name,total_calls,average from top_kernels where name like '%k_ct_coords%' or name like '%k_levels%' or name like '%k_ilu0%'"):
    print('$tag', r[0][:60], r[1], round(r[2]/1000,1), 'ms')
PY
