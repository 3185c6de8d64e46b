# round 4, call 25: longer placement searches (RAMD_PLACE_DRAWS) against the default 8, fresh processes alternating
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04y
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 100 --warmup 10"
for rep in 1 2 3 4 5 6; do
  for v in 8 24; do
    RAMD_PLACE_DRAWS=$v timeout 600 python $R/bench.py $B 2> /dev/null | grep '^{' > $O/line_d${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04y/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms'], 'placement_s', d['placement_s'])
    except Exception as e: print(f, e)
PY
