# round 4, call 12: un-profiled bench lines of the other configurations with the final code; why the vendor backend's GMRES+ILU(0) fails at 512^3
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
run() { name=$1; shift; timeout 900 python $R/bench.py "$@" 2> $O/line_$name.err | grep '^{' > $O/line_$name.json; echo "line $name rc=$?"; }
run ell $B --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10
run hyb $B --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10
run bicgstab_csr $B --solver bicgstab --precond mcsgs --steps 60 --warmup 10
run mixed $B --solver mixed --steps 30 --warmup 3
RAMD_CSR_PAT=0 run mixed_columns_read $B --solver mixed --steps 30 --warmup 3
run gmres $B --solver gmres --precond ilu0 --steps 60 --warmup 10
run shell --no-reference-gpu --no-extras --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10
run cg256 $B --grid 256 --steps 200 --warmup 20
run global1 $B --force-global --steps 100 --warmup 10
for N in 512 256; do
  timeout 600 $R/oracle/_ref/ref_probe bench $N 20 0 1 gmres ilu0 > $O/ref_gmres_$N.out 2> $O/ref_gmres_$N.err; echo "ref gmres ilu0 $N rc=$?"
  tail -3 $O/ref_gmres_$N.err; tail -1 $O/ref_gmres_$N.out | cut -c1-300
done
