# the measurements behind profiles/r05_* on the final tree of the round: rocprofv3 passes of the legs whose kernels changed since
# tools/r05_runs/final.sh ran, all bench lines again
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05final2
mkdir -p $O
cd $R
bash tools/profile_r05.sh cg gmres shell shell_rcm bicgstab_rb mixed > $O/profile.log 2>&1
cd /tmp; export TMPDIR=/tmp
line() { name=$1; shift; timeout 900 python $R/bench.py "$@" 2> $O/$name.err | grep '^{' > $O/bench_line_$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/bench_line_$name.json'));print(d['value'],d['unit'])" 2>/dev/null)"; }
line cg
line gmres_ilu0 --solver gmres --precond ilu0 --steps 60 --warmup 10
line bicgstab_mcsgs --solver bicgstab --precond mcsgs --steps 60 --warmup 10
line ell --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10
line hyb --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10
line mixed --solver mixed --steps 30 --warmup 3
line shell --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10
for k in rcm delaunay random; do line shell_$k --matrix shell --shell-variant $k --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu; done
line cg_256 --grid 256 --steps 200 --warmup 20
line gmres_ilu0_256 --grid 256 --solver gmres --precond ilu0 --steps 60 --warmup 10
line global_1rank --force-global
