# rocprofv3 passes behind profiles/r04_* (run on the GPU box from the repo root; then: python tools/prof_summary.py r04)
# kernel traces and counter passes are separate runs (gpurun refuses --pmc together with the runtime trace domains)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
B="--no-cpu-baseline --no-reference-gpu --no-extras"
kt() { # name, bench args...
  name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $P/kt_$name -o bench -- python $R/bench.py $B "$@" > $P/bench_kt_$name.json 2> $P/kt_$name.err
  grep '^{' $P/bench_kt_$name.json > $P/bench_kt_$name.json.tmp; mv $P/bench_kt_$name.json.tmp $P/bench_kt_$name.json
  echo "kernel trace $name rc=$?"
}
pmc() { # name, counter, bench args...
  name=$1; ctr=$2; shift; shift
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace -d $P/${ctr}_$name -o bench -- python $R/bench.py $B "$@" > /dev/null 2> $P/${ctr}_$name.err
  echo "pmc $ctr $name rc=$?"
}
kt cg --steps 100 --warmup 10
kt gmres --solver gmres --precond ilu0 --steps 60 --warmup 10
kt shell --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10
kt bicgstab --solver bicgstab --precond mcsgs --steps 60 --warmup 10
kt ell --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10
kt hyb --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10
kt mixed --solver mixed --steps 30 --warmup 3
for c in FETCH_SIZE WRITE_SIZE; do
  pmc cg $c --steps 20 --warmup 2
  pmc gmres $c --solver gmres --precond ilu0 --steps 20 --warmup 2
  pmc shell $c --matrix shell --solver gmres --precond ilu0 --steps 20 --warmup 2
  pmc bicgstab $c --solver bicgstab --precond mcsgs --steps 20 --warmup 2
  pmc ell $c --format ell --solver bicgstab --precond mcsgs --steps 20 --warmup 2
  pmc mixed $c --solver mixed --steps 10 --warmup 2
  # calibration of the counters on known byte counts (16 / 8 / 4 / 1 bytes per lane; tools/membench.hip)
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d $P/${c}_calib -o bench -- $R/tools/_bin/membench calib > $P/${c}_calib.log 2>&1
  echo "pmc $c calib rc=$?"
done
# the sqlite files are large: the text summaries are made on the box and only they travel back
cd $R && python tools/prof_summary.py r04 box
ls $P
