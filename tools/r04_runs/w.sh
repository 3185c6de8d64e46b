# round 4, call 23: config 4 (ELL / HYB interiors) with k_ell2: bench lines, kernel stats, counters; the whole GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04w
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
run() { name=$1; shift; timeout 900 python $R/bench.py "$@" 2> /dev/null | grep '^{' > $O/line_$name.json; echo "line $name rc=$?"; }
run ell $B --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10
run hyb $B --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10
run cgell $B --format ell --steps 100 --warmup 10
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats -d $P/kt_ell -o bench -- python $R/bench.py $B --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10 2> /dev/null | grep '^{' > $P/bench_kt_ell.json
timeout 900 rocprofv3 --kernel-trace --stats -d $P/kt_hyb -o bench -- python $R/bench.py $B --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10 2> /dev/null | grep '^{' > $P/bench_kt_hyb.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace -d $P/${c}_ell -o bench -- python $R/bench.py $B --format ell --solver bicgstab --precond mcsgs --steps 20 --warmup 2 > /dev/null 2>&1
done
cd $R && python tools/prof_summary.py r04 box > /dev/null; ls gpurun_out/prof_txt | head -20
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
