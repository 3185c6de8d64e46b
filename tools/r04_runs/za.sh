# round 4, call 27: CG headline with the matrix arrays carved from an arena (and the work vectors not), alternating
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04za
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 100 --warmup 10"
for rep in 1 2 3 4; do
  for cfg in "def RAMD_ALLOC_ARENA=0" "all RAMD_ALLOC_ARENA=1" "mat RAMD_ALLOC_ARENA=1 RAMD_ARENA_MIN_MB=2048"; do
    set -- $cfg; name=$1; shift
    env "$@" timeout 600 python $R/bench.py $B 2> /dev/null | grep '^{' > $O/line_${name}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04za/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['roofline_columns_read']['avg_ms'], d['kernels']['vector_updates']['avg_ms'], 'placement_s', d['placement_s'])
    except Exception as e: print(f, e)
PY
