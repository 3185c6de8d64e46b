# the measurements behind profiles/r05_*: rocprofv3 passes of every leg, the bench lines, ten fresh-process lines after a heavy job
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05final
mkdir -p $O
cd $R
bash tools/profile_r05.sh cg gmres shell bicgstab ell hyb mixed calib > $O/profile.log 2>&1
cd /tmp; export TMPDIR=/tmp
line() { name=$1; shift; timeout 900 python $R/bench.py "$@" 2> $O/$name.err | grep '^{' > $O/bench_line_$name.json; echo "$name rc=$?"; }
line cg
line gmres_ilu0 --solver gmres --precond ilu0 --steps 60 --warmup 10
line bicgstab_mcsgs --solver bicgstab --precond mcsgs --steps 60 --warmup 10
line ell --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10
line hyb --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10
line mixed --solver mixed --steps 30 --warmup 3
line shell --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10
line cg_256 --grid 256 --steps 200 --warmup 20
line gmres_ilu0_256 --grid 256 --solver gmres --precond ilu0 --steps 60 --warmup 10
line global_1rank --force-global
# ten fresh processes right after a heavy job
cd $R; timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_box_tiles_forced.py -m gpu -q > /dev/null 2>&1; cd /tmp
for i in 0 1 2 3 4 5 6 7 8 9; do
  timeout 400 python $R/bench.py --no-cpu-baseline --no-reference-gpu --no-extras --steps 60 --warmup 10 2>/dev/null | grep '^{' > $O/repeat_$i.json
done
python3 - <<PY
import json,glob,os
O="$O"
for f in sorted(glob.glob(O+"/repeat_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["value"], "it/s  spmv", d["roofline"]["avg_ms"], "updates", d["kernels"]["vector_updates"]["avg_ms"], "placement_s", d["placement_s"])
    except Exception as e: print(f, e)
PY
