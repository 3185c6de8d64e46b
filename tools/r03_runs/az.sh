#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03az; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RAMD_ROCTX=1 timeout 600 rocprofv3 --marker-trace --kernel-trace -d $O/roctx -o m -- python $R/bench.py --grid 128 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras > $O/b.json 2> $O/b.err; echo "rc=$?"
python - <<PY
import sqlite3,glob
for f in glob.glob('$O/roctx/*.db'):
    cur=sqlite3.connect(f).cursor()
    try:
        print(cur.execute("select name, count(*) from regions group by name order by 2 desc limit 10").fetchall())
    except Exception as e:
        print('regions query failed', e)
        print([r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")][:40])
PY
