# after the last kernel changes of the round: the RCM leg's passes and the three variant lines again; then the whole GPU suite with the
# sync-free grouped form forced wherever it applies (a shake-out, not part of the suite: tests that assert another form fail by design)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05final3
mkdir -p $O
cd $R
bash tools/profile_r05.sh shell_rcm > $O/profile.log 2>&1
cd /tmp; export TMPDIR=/tmp
line() { name=$1; shift; timeout 900 python $R/bench.py "$@" 2> $O/$name.err | grep '^{' > $O/bench_line_$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/bench_line_$name.json'));print(d['value'],d['unit'],d['roofline']['min_ms'],d['roofline']['max_ms'])" 2>/dev/null)"; }
for k in rcm delaunay random; do line shell_$k --matrix shell --shell-variant $k --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu; done
cd $R
RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 RAMD_TRSV_BAND=0 timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_box_tiles_forced.py --deselect tests/test_gpu_lattice.py --deselect tests/test_gpu_syncfree.py > $O/forced_suite.log 2>&1; echo "forced suite rc=$?"; tail -40 $O/forced_suite.log | cut -c1-200
