# round 4, call 24: the pair kernel of the ELL product with the columns read (four slots per batch) against k_ell
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04x
mkdir -p $O
cd $R
RAMD_ELL2=2 RAMD_CSR_PAT=0 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -m gpu -q -x -k "(ell or hyb or format or convert or golden or apply) and not fresh_process and not forced" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --format ell --steps 100 --warmup 10"
for rep in 1 2 3; do
  for v in 0 2; do
    RAMD_CSR_PAT=0 RAMD_ELL2=$v timeout 600 python $R/bench.py $B 2>/dev/null | grep '^{' > $O/line_e${v}_$rep.json
  done
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04x/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['roofline']['frac'])
    except Exception as e: print(f, e)
PY
