# round 4, call 4: where the cycles of the 512^3 triangular solves go (phase timers), masked publication without dependency waits
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04d
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mp in 0 1; do
  RAMD_TRSV_MASKPUB=$mp TAG=mp$mp timeout 300 python $R/tools/trsv_time.py poisson 512 >> $O/time.log 2>&1
  RAMD_TRSV_NOFILL=1 RAMD_TRSV_MASKPUB=$mp TAG=nofill_mp$mp timeout 300 python $R/tools/trsv_time.py poisson 512 >> $O/time.log 2>&1
done
RAMD_TRSV_PROF=1 TAG=prof timeout 300 python $R/tools/trsv_time.py poisson 512 > $O/prof.log 2>&1
RAMD_TRSV_PROF=1 RAMD_TRSV_NOFILL=1 TAG=prof_nofill timeout 300 python $R/tools/trsv_time.py poisson 512 > $O/prof_nofill.log 2>&1
B="--no-cpu-baseline --no-reference-gpu --no-extras --solver gmres --precond ilu0 --steps 60 --warmup 10"
timeout 900 python $R/bench.py --matrix shell $B > $O/line_shell.json 2> $O/line_shell.err; echo "shell rc=$?"
tail -3 $O/line_shell.err
cat $O/time.log | grep -v "^rocal"
tail -4 $O/prof.log | cut -c1-400
