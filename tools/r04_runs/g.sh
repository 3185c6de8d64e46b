# round 4, call 7: the 35-entry-row product with the look-ahead of k_csr_w4 (PF) against the old form, alternating
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_forced and W4" > $O/pytest.log 2>&1
echo "pytest w4 rc=$?"; tail -2 $O/pytest.log
timeout 600 python -m pytest tests/test_gpu_shell.py -m gpu -q -x > $O/pytest_shell.log 2>&1
echo "pytest shell rc=$?"; tail -2 $O/pytest_shell.log
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for pf in 0 1; do
    RAMD_CSR_W4_PF=$pf TAG=pf$pf timeout 300 python $R/tools/spmv_shell.py 549 2>&1 | grep "^shell" >> $O/spmv.log
  done
done
RAMD_CSR_W4=0 TAG=tr timeout 300 python $R/tools/spmv_shell.py 549 2>&1 | grep "^shell" >> $O/spmv.log
cat $O/spmv.log
# ... and the row-pattern product with more waves per SIMD (late request of the second block / 7-8 waves asked of the compiler)
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_forced and LATE" > $O/pytest_late.log 2>&1
echo "pytest late rc=$?"; tail -2 $O/pytest_late.log
cd /tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 100 --warmup 10"
for rep in 1 2 3; do
for cfg in "base RAMD_CSR_PAT2_LATE=0" "late8 RAMD_CSR_PAT2_LATE=8" "late7 RAMD_CSR_PAT2_LATE=7" "w7 RAMD_CSR_PAT2_LATE=17"; do
  set -- $cfg; name=$1; shift
  env "$@" timeout 600 python $R/bench.py $B > $O/line_${name}_$rep.json 2> $O/line_${name}_$rep.err; echo "line $name $rep rc=$?"
done; done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04g/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_ms'], {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, e)
PY
