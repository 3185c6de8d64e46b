# round 4, call 21: ELL product with two rows per thread (k_ell2) against k_ell, alternating; format suites
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04u
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_forced and ELL2" > $O/pytest_var.log 2>&1
echo "pytest variants rc=$?"; tail -2 $O/pytest_var.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py tests/test_gpu_full_size.py tests/test_gpu_global_full_size.py -m gpu -q -x -k "(ell or hyb or format or convert or config or full) and not fresh_process and not forced and not soak" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for rep in 1 2 3; do
  for v in 0 1; do
    RAMD_ELL2=$v timeout 600 python $R/bench.py $B --format ell --steps 100 --warmup 10 2>/dev/null | grep '^{' > $O/line_cgell_e${v}_$rep.json
    RAMD_ELL2=$v timeout 600 python $R/bench.py $B --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10 2>/dev/null | grep '^{' > $O/line_hyb_e${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04u/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['roofline']['frac'], 'colsread', d.get('roofline_columns_read',{}).get('avg_ms'), d.get('roofline_columns_read',{}).get('frac'))
    except Exception as e: print(f, e)
PY
