# round 4, call 54: the end-of-round sequence on the final tree: GPU suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zw
mkdir -p $O
cd $R
S=$SECONDS
timeout 3300 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1
echo "suite rc=$? $((SECONDS-S)) s"; tail -3 $O/suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
S=$SECONDS
timeout 1500 python bench.py > $O/bench.out 2> $O/bench.err; echo "bench rc=$? $((SECONDS-S)) s"
grep '^{' $O/bench.out > $O/bench_line.json
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["cpu_baseline"]["value"], d.get("reference_gpu",{}).get("value"), d["placement_s"])
for e in d.get("extras",[]): print(e.get("metric","")[:50], e.get("value"))
PY
