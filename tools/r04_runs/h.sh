# round 4, call 8: the first stage's sentinel fill folded into the second stage's read of w (A/B), triangular-solve suites
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04h
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_shell.py tests/test_gpu_box_tiles_forced.py tests/test_gpu_edge_cases.py -m gpu -q -x -k "lusolve or trisolve or ilu or sgs or ic or shell or box_tiles or solvers_vs_golden or llsolve or lsolve or usolve or preconditioner" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver gmres --precond ilu0 --steps 60 --warmup 10"
for rep in 1 2 3; do
for v in 0 1; do
  RAMD_TRSV_REFILL=$v timeout 600 python $R/bench.py $B > $O/line_rf${v}_$rep.json 2> $O/line_rf${v}_$rep.err; echo "line refill=$v $rep rc=$?"
done; done
for v in 0 1; do
  RAMD_TRSV_REFILL=$v TAG=refill$v timeout 300 python $R/tools/trsv_time.py poisson 512 2>&1 | grep "^poisson" >> $O/time.log
  RAMD_TRSV_REFILL=$v TAG=refill$v timeout 300 python $R/tools/trsv_time.py shell 549 2>&1 | grep "^shell" >> $O/time.log
done
cat $O/time.log
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04h/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_ms'], {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, e)
PY
