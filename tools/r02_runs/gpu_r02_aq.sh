#!/bin/bash
mkdir -p gpurun_out/r02aq
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02aq/kt -o b -- python $R/bench.py --matrix shell --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu --no-extras > $R/gpurun_out/r02aq/b.json 2> $R/gpurun_out/r02aq/b.err
cd $R; python - <<PY
import sqlite3
c=sqlite3.connect('gpurun_out/r02aq/kt/b_results.db').cursor()
for r in c.execute("select name,total_calls,average,total_duration from top_kernels order by total_duration desc limit 14"):
    print(r[0][:70].replace('void ramd::',''), r[1], round(r[2]/1000,2), 'ms avg', round(r[3]/1e6,1),'ms total')
PY
