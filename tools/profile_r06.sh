# rocprofv3 passes behind profiles/r06_* (run on the GPU box from the repo root; then the text summaries are merged back under
# gpurun_out/prof_txt and copied to profiles/).  Kernel traces and counter passes are separate runs (gpurun refuses --pmc
# together with the runtime trace domains).   bash tools/profile_r06.sh [leg ...]   legs: cg gmres shell shell_rcm bicgstab_rb mixed lap27_cg lap27_gmres lap27_bicgstab lap27_ell lap27_hyb
# (bicgstab / ell / hyb: the colour sweeps, RAMD_MC_RB=0; *_rb: the default, the one-pass red-black form)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
B="--no-cpu-baseline --no-reference-gpu --no-extras"
LEGS=${@:-cg gmres shell_rcm bicgstab_rb lap27_cg lap27_gmres lap27_bicgstab lap27_ell}
args_of() {
  case $1 in
    cg) echo "";;
    gmres) echo "--solver gmres --precond ilu0";;
    shell) echo "--matrix shell --solver gmres --precond ilu0";;
    shell_rcm) echo "--matrix shell --shell-variant rcm --solver gmres --precond ilu0";;
    bicgstab) echo "--solver bicgstab --precond mcsgs";;
    ell) echo "--format ell --solver bicgstab --precond mcsgs";;
    hyb) echo "--format hyb --solver bicgstab --precond mcsgs";;
    bicgstab_rb) echo "--solver bicgstab --precond mcsgs";;
    ell_rb) echo "--format ell --solver bicgstab --precond mcsgs";;
    hyb_rb) echo "--format hyb --solver bicgstab --precond mcsgs";;
    mixed) echo "--solver mixed";;
    lap27_cg) echo "--matrix lap27 --grid 256";;
    lap27_gmres) echo "--matrix lap27 --grid 256 --solver gmres --precond ilu0";;
    lap27_bicgstab) echo "--matrix lap27 --grid 256 --solver bicgstab --precond mcsgs";;
    lap27_ell) echo "--matrix lap27 --grid 256 --format ell";;
    lap27_hyb) echo "--matrix lap27 --grid 256 --format hyb";;
  esac
}
env_of() { case $1 in bicgstab|ell|hyb) export RAMD_MC_RB=0;; *) unset RAMD_MC_RB;; esac; }
for leg in $LEGS; do
  [ $leg = calib ] && continue
  env_of $leg
  st="--steps 60 --warmup 10"; [ $leg = cg ] && st="--steps 100 --warmup 10"; [ $leg = mixed ] && st="--steps 30 --warmup 3"
  timeout 900 rocprofv3 --kernel-trace --stats -d $P/kt_$leg -o bench -- python $R/bench.py $B $(args_of $leg) $st > $P/bench_kt_$leg.json 2> $P/kt_$leg.err
  grep '^{' $P/bench_kt_$leg.json > $P/bench_kt_$leg.json.tmp; mv $P/bench_kt_$leg.json.tmp $P/bench_kt_$leg.json
  echo "kernel trace $leg done"
done
for c in FETCH_SIZE WRITE_SIZE; do
  for leg in $LEGS; do
    if [ $leg = calib ]; then
      # calibration of the counters on known byte counts (16 / 8 / 4 / 1 bytes per lane; tools/membench.hip)
      timeout 300 rocprofv3 --pmc $c --kernel-trace -d $P/${c}_calib -o bench -- $R/tools/_bin/membench calib > $P/${c}_calib.log 2>&1
    else
      env_of $leg
      st="--steps 20 --warmup 2"; [ $leg = mixed ] && st="--steps 10 --warmup 2"
      timeout 900 rocprofv3 --pmc $c --kernel-trace -d $P/${c}_$leg -o bench -- python $R/bench.py $B $(args_of $leg) $st > /dev/null 2> $P/${c}_$leg.err
    fi
    echo "pmc $c $leg done"
  done
done
# the sqlite files are large: the text summaries are made on the box and only they travel back
cd $R && python tools/prof_summary.py r06 box
ls $P
