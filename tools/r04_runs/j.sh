# round 4, call 10: ELL / colour sweeps with the first batch of values requested before the dictionary barrier
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04j
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -m gpu -q -x -k "(ell or hyb or spmv or apply or mcsgs or mcgs or mcilu or multicolo or golden or convert) and not fresh_process and not forced" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_forced and (PAT=1 or PAT=0) and not W4 and not Q4 and not XL and not GRP and not COL2 and not NORP and not PIPE" > $O/pytest_var.log 2>&1
echo "pytest variants rc=$?"; tail -2 $O/pytest_var.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for rep in 1 2; do
  timeout 600 python $R/bench.py $B --solver bicgstab --precond mcsgs --format ell --steps 60 --warmup 10 > $O/line_ell_$rep.json 2> $O/line_ell_$rep.err; echo "ell $rep rc=$?"
  timeout 600 python $R/bench.py $B --solver bicgstab --precond mcsgs --steps 60 --warmup 10 > $O/line_csr_$rep.json 2> $O/line_csr_$rep.err; echo "csr $rep rc=$?"
  timeout 600 python $R/bench.py $B --format ell --steps 100 --warmup 10 > $O/line_cgell_$rep.json 2> $O/line_cgell_$rep.err; echo "cgell $rep rc=$?"
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04j/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d.get('roofline_columns_read',{}).get('avg_ms'), {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, e)
PY
