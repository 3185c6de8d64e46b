# round 4, call 20: placement draws with spacers between them (RAMD_ALLOC_SPACER) against plain draws, fresh processes alternating
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 100 --warmup 10"
for rep in 1 2 3 4 5 6; do
  for v in 0 1; do
    RAMD_ALLOC_SPACER=$v timeout 600 python $R/bench.py $B 2> /dev/null | grep '^{' > $O/line_sp${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04t/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms'], 'placement_s', d['placement_s'], 'build', d['build_s'])
    except Exception as e: print(f, e)
PY
