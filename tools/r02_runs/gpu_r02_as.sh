#!/bin/bash
mkdir -p gpurun_out/r02as
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02as/kt -o b -- python $R/bench.py --solver bicgstab --precond mcsgs --format ell --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras > $R/gpurun_out/r02as/b.json 2> $R/gpurun_out/r02as/b.err
cd $R; python - <<PY
import sqlite3, json
c=sqlite3.connect('gpurun_out/r02as/kt/b_results.db').cursor()
for r in c.execute("select name,total_calls,average,total_duration from top_kernels order by total_duration desc limit 16"):
    print(r[0][:90].replace('void ramd::',''), r[1], round(r[2]/1000,3), 'ms avg', round(r[3]/1e6,1),'ms total')
d=json.loads(open('gpurun_out/r02as/b.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])
PY
