#!/bin/bash
O=gpurun_out/r03ck; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4 5 6 7 8 9 10; do
timeout 600 python bench.py --steps 100 --warmup 10 $B > $O/cg_$i.json 2> $O/cg_$i.err
python - $O/cg_$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print('run', d['value'], d['ms_per_step'], 'spmv', d['roofline']['avg_ms'], d['roofline']['frac'], 'placement', d.get('placement_s'))
PY
done
R=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/kt_cg -o bench -- python $R/bench.py $B --steps 100 --warmup 10 > $R/$O/bench_kt_cg.json 2> $R/$O/kt_cg.err; echo "kt rc=$?")
python - <<'PY'
import sqlite3,glob
for db in glob.glob('gpurun_out/r03ck/kt_cg/**/bench_results.db', recursive=True):
    for r in sqlite3.connect(db).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 10").fetchall():
        print("%s | %d | %.1f | %.3f | %.2f" % (r[0].replace("void ramd::","ramd::")[:100], r[1], r[2], r[3], r[4]))
PY
