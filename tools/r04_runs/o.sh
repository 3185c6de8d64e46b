# round 4, call 15: 256^3 lines next to the vendor backend (the size at which the vendor ILU(0) runs)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04o
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 900 python $R/bench.py "$@" 2> $O/line_$name.err | grep '^{' > $O/line_$name.json; echo "line $name rc=$?"; }
run gmres256 --no-cpu-baseline --no-extras --grid 256 --solver gmres --precond ilu0 --steps 60 --warmup 10
run bicgstab256 --no-cpu-baseline --no-extras --grid 256 --solver bicgstab --precond mcsgs --steps 100 --warmup 10
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04o/line_*.json')):
    d=json.load(open(f)); print(os.path.basename(f), d['value'], d['ms_per_step'], json.dumps(d.get('reference_gpu'))[:300])
PY
