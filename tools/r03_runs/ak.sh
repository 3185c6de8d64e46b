#!/bin/bash
# final measurement pass of round 3: kernel traces (the counter passes of profiles/r03_pmc_* stay: the solve kernels did not change),
# the bench lines of record, CG repeats in fresh processes
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03ak; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof; mkdir -p $P
B="--no-cpu-baseline --no-reference-gpu --no-extras"
kt() { name=$1; shift
  rm -rf $P/kt_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d $P/kt_$name -o bench -- python $R/bench.py $B "$@" > $P/bench_kt_$name.json 2> $P/kt_$name.err
  grep '^{' $P/bench_kt_$name.json > $P/bench_kt_$name.json.tmp; mv $P/bench_kt_$name.json.tmp $P/bench_kt_$name.json
  echo "kernel trace $name done"
}
kt cg --steps 100 --warmup 10
kt gmres --solver gmres --precond ilu0 --steps 60 --warmup 10
kt shell --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10
kt bicgstab --solver bicgstab --precond mcsgs --steps 60 --warmup 10
# drop the bulky per-dispatch traces, keep the databases prof_summary.py reads
find $P -name "*.csv" -size +2M -delete
cd $R
timeout 1200 python bench.py > $O/cg_default.json 2> $O/cg_default.err; echo "default rc=$?"
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu > $O/gmres.json 2> $O/gmres.err; echo "gmres rc=$?"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu > $O/shell.json 2> $O/shell.err; echo "shell rc=$?"
timeout 900 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/bicgstab.json 2> $O/bicgstab.err; echo "bicgstab rc=$?"
timeout 900 python bench.py --force-global --steps 100 --warmup 10 > $O/global1.json 2> $O/global1.err; echo "global1 rc=$?"
timeout 900 python bench.py --grid 256 --precond global-saamg --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras > $O/gsaamg256.json 2> $O/gsaamg256.err; echo "global saamg rc=$?"; tail -3 $O/gsaamg256.err
for i in 1 2 3 4 5 6 7 8; do
  timeout 600 python bench.py --steps 200 --warmup 20 $B > $O/a$i.json 2> $O/a$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03ak/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], 'build', d.get('build_s'), (d.get('roofline') or {}).get('avg_ms'), (d.get('roofline') or {}).get('frac'), {k:(v['avg_ms'],v['frac']) for k,v in (d.get('kernels') or {}).items()}, (d.get('cpu_baseline') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
