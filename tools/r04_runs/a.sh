# round 4, call 1: the GPU suite with this round's host changes, then configs 4 / 5 with this round's code (bench lines,
# kernel stats, counters) and counter passes over the colour sweeps
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
kt() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o bench -- python $R/bench.py $B "$@" > $O/bench_kt_$name.json 2> $O/kt_$name.err
  echo "kernel trace $name rc=$?"; }
pmc() { name=$1; ctr=$2; shift; shift
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_${name}_$(echo $ctr | tr ' ' '_' | cut -c1-40) -o bench -- python $R/bench.py $B "$@" > /dev/null 2> $O/pmc_${name}.err
  echo "pmc $ctr $name rc=$?"; }
# plain lines first (un-profiled numbers)
for cfg in "mixed --solver mixed --steps 30 --warmup 3" "ell --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10" "hyb --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10" "cgell --format ell --steps 100 --warmup 10"; do
  set -- $cfg; name=$1; shift
  timeout 600 python $R/bench.py $B "$@" > $O/line_$name.json 2> $O/line_$name.err; echo "line $name rc=$?"
done
kt mixed --solver mixed --steps 30 --warmup 3
kt ell --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10
kt hyb --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10
for c in FETCH_SIZE WRITE_SIZE; do
  pmc mixed $c --solver mixed --steps 10 --warmup 2
  pmc ell $c --format ell --solver bicgstab --precond mcsgs --steps 20 --warmup 2
done
# where do the colour sweeps' extra bytes come from: L2 hit/miss, request mix, L1 side
pmc ell "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" --format ell --solver bicgstab --precond mcsgs --steps 20 --warmup 2
pmc ell "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" --format ell --solver bicgstab --precond mcsgs --steps 20 --warmup 2
pmc ell "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum" --format ell --solver bicgstab --precond mcsgs --steps 20 --warmup 2

python $R/tools/db_summary.py $O; ls $O
