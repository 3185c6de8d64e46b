# round 4, call 28: the whole GPU suite and the default bench line with matrix arrays in arenas by default
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zb
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  timeout 1500 python $R/bench.py --steps 20 --warmup 5 2> $O/bench_$rep.err | grep '^{' > $O/bench_$rep.json; echo "bench $rep rc=$?"
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04zb/bench_*.json')):
    d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['frac'], d['roofline_columns_read']['frac'], d['kernels']['vector_updates']['avg_ms'], {k:(v.get('iters_per_s'), v.get('roofline',{}).get('avg_ms')) for k,v in d['extras'].items()})
PY
