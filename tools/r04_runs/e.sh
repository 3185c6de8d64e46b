# round 4, call 5: colour sweeps -- block orders (round robin / contiguous / band tiles of several widths) with the prologue that
# requests the row data before the dictionary is staged; alternating runs
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_solvers.py -m gpu -q -x -k "mcsgs or mcgs or mcilu or multicolo or preconditioner_apply" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver bicgstab --precond mcsgs --steps 60 --warmup 10"
for rep in 1 2; do
for cfg in "rr RAMD_MC_XCD=0" "lin RAMD_MC_XCD=2" "w32 RAMD_MC_XCD=1" "w64 RAMD_MC_XCD=1 RAMD_MC_BANDW=64" "w128 RAMD_MC_XCD=1 RAMD_MC_BANDW=128" "w512 RAMD_MC_XCD=1 RAMD_MC_BANDW=512" "w8 RAMD_MC_XCD=1 RAMD_MC_BANDW=8"; do
  set -- $cfg; name=$1; shift
  env "$@" timeout 600 python $R/bench.py $B > $O/line_${name}_$rep.json 2> $O/line_${name}_$rep.err; echo "line $name $rep rc=$?"
done; done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04e/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'])
    except Exception as e: print(f, e)
PY
