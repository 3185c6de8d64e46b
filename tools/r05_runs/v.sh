# k_trsv_sf integrated: the new tests, then GMRES(30)+ILU(0) on the config-3 class in all four numberings
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05v
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_syncfree.py tests/test_gpu_box_tiles_forced.py -m gpu -x -q --durations=8 -k "sync_free or band or row_groups or rcm_numbered" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -14 $O/tests.log
cd /tmp; export TMPDIR=/tmp
for k in rcm delaunay random lex; do
  timeout 900 python $R/bench.py --matrix shell --shell-variant $k --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu 2> $O/$k.err | grep '^{' > $O/bench_line_shell_$k.json
  python3 - <<PY
import json
d=json.load(open("$O/bench_line_shell_$k.json"))
tp=d.get("tri_plan",{})
print("$k", d["value"], "it/s", d["ms_per_step"], "ms/it |", {k:(v.get("form","")[:40], v.get("dependency_levels")) for k,v in tp.items()} if isinstance(tp,dict) else tp, "|", {k:v for k,v in d.get("kernels",{}).items() if "tri" in k.lower() or "lu" in k.lower()})
PY
done
