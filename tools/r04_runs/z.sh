# round 4, call 26: GMRES(30)+ILU(0) with every big block carved from arenas (one arena = one placement class) against the default
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver gmres --precond ilu0 --steps 60 --warmup 10"
for rep in 1 2 3; do
  for v in 0 1; do
    RAMD_ALLOC_ARENA=$v timeout 600 python $R/bench.py $B 2> /dev/null | grep '^{' > $O/line_a${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04z/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_ms'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
    except Exception as e: print(f, e)
PY
