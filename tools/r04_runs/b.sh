# round 4, call 2: colour sweeps -- XCD/band-aware block order and output pairs, A/B in alternating runs + counters; GPU suite
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04b
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver bicgstab --precond mcsgs --steps 60 --warmup 10"
for rep in 1 2; do
for cfg in "00 RAMD_MC_XCD=0 RAMD_MC_PAIR=0" "10 RAMD_MC_XCD=1 RAMD_MC_PAIR=0" "01 RAMD_MC_XCD=0 RAMD_MC_PAIR=1" "11 RAMD_MC_XCD=1 RAMD_MC_PAIR=1"; do
  set -- $cfg; name=$1; shift
  env "$@" timeout 600 python $R/bench.py $B > $O/line_${name}_$rep.json 2> $O/line_${name}_$rep.err; echo "line $name $rep rc=$?"
done; done
pmc() { name=$1; ctr=$2; shift; shift
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_${name}_$(echo $ctr | tr ' ' '_' | cut -c1-40) -o bench -- python $R/bench.py $B "$@" > /dev/null 2> $O/pmc_${name}.err
  echo "pmc $ctr $name rc=$?"; }
pmc new FETCH_SIZE --steps 20 --warmup 2
pmc new WRITE_SIZE --steps 20 --warmup 2
python $R/tools/db_summary.py $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
