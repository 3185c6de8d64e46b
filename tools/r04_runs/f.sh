# round 4, call 6: persistent row-pattern product (k_csr_pat3) against k_csr_pat2 in alternating runs; colour sweeps with 16-byte value pairs
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04f
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "variants_forced and PAT3" > $O/pytest.log 2>&1
echo "pytest pat3 rc=$?"; tail -2 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_solvers.py -m gpu -q -x -k "mcsgs or mcgs or mcilu or multicolo or preconditioner_apply" > $O/pytest_mc.log 2>&1
echo "pytest mc rc=$?"; tail -2 $O/pytest_mc.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 100 --warmup 10"
for rep in 1 2 3; do
for cfg in "p2 RAMD_CSR_PAT3=0" "p3 RAMD_CSR_PAT3=1" "p3w2 RAMD_CSR_PAT3=1 RAMD_CSR_PAT3_WGS=2"; do
  set -- $cfg; name=$1; shift
  env "$@" timeout 600 python $R/bench.py $B > $O/line_${name}_$rep.json 2> $O/line_${name}_$rep.err; echo "line $name $rep rc=$?"
done; done
BB="--no-cpu-baseline --no-reference-gpu --no-extras --solver bicgstab --precond mcsgs --steps 60 --warmup 10"
for rep in 1 2; do
  timeout 600 python $R/bench.py $BB > $O/line_mc_$rep.json 2> $O/line_mc_$rep.err; echo "line mc $rep rc=$?"
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04f/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], d['roofline']['avg_ms'], {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(f, e)
PY
