# round 4, call 34: the distributed aggregation through the C ABI against the oracle's P-way mode and the goldens;
# then the coupled AMG at a larger size over 4 ranks on one device (Build time of the host-side bookkeeping)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zg
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -k "distributed_pmis" > $O/agg.log 2>&1
tail -30 $O/agg.log
for pc in global-uaamg global-saamg; do
  for g in 1 4; do
    S=$SECONDS
    timeout 900 python bench.py --gpus $g --transport callback --grid 128 --precond $pc --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras 2>$O/${pc}_$g.err | grep '^{' > $O/${pc}_$g.json
    echo "$pc gpus=$g wall=$((SECONDS-S))s $(python - <<PY
import json
try:
    d=json.loads(open("$O/${pc}_$g.json").read())
    print("it/s", d["value"], "build_s", d.get("build_s"), "final", d.get("final_residual"), d.get("config",{}).get("workload","")[:60])
except Exception as e:
    print("no line", e)
PY
)"
    tail -3 $O/${pc}_$g.err
  done
done
