# round 4, call 22: colour sweeps with two rows per thread (k_mc_sweep2) against k_mc_sweep, alternating; multi-colour suites
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04v
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_solvers.py -m gpu -q -x -k "mcsgs or mcgs or mcilu or multicolo or preconditioner_apply or forced" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
RAMD_MC2=1 RAMD_CSR_PAT=1 timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_kernels.py -m gpu -q -x -k "(mcsgs or mcgs or mcilu or multicolo or preconditioner_apply or golden) and not forced and not fresh" > $O/pytest_mc2.log 2>&1
echo "pytest mc2 rc=$?"; tail -2 $O/pytest_mc2.log
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_global_full_size.py -m gpu -q -x -k "not soak" > $O/pytest_full.log 2>&1
echo "pytest full rc=$?"; tail -2 $O/pytest_full.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver bicgstab --precond mcsgs --steps 60 --warmup 10"
for rep in 1 2 3; do
  for v in 0 1; do
    RAMD_MC2=$v timeout 600 python $R/bench.py $B 2>/dev/null | grep '^{' > $O/line_mc2${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04v/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_ms'], {k:v['avg_ms'] for k,v in d['kernels'].items()})
    except Exception as e: print(f, e)
PY
