#!/bin/bash
O=gpurun_out/r03ca; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2; do
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 $B > $O/g$i.json 2> $O/g$i.err
done
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 $B > $O/s1.json 2> $O/s1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03ca/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], 'build', d.get('build_s'), d['roofline'])
PY
R=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/kt_gmres -o bench -- python $R/bench.py $B --solver gmres --precond ilu0 --steps 60 --warmup 10 > $R/$O/bench_kt_gmres.json 2> $R/$O/kt_gmres.err; echo "kt rc=$?")
python - <<'PY'
import sqlite3,glob
for db in glob.glob('gpurun_out/r03ca/kt_gmres/**/bench_results.db', recursive=True):
    for r in sqlite3.connect(db).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 14").fetchall():
        print("%s | %d | %.1f | %.3f | %.2f" % (r[0].replace("void ramd::","ramd::")[:110], r[1], r[2], r[3], r[4]))
PY
bash tools/pmc_trsv.sh $O/pmc 2>&1 | tail -8
