# the BASELINE.json configurations (and the widened preconditioners) through bench.py on one GPU
R=${GRAFT_REPO_ROOT:-.}
cd $R
O=gpurun_out/configs
mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras "$@" 2>$O/$name.err | grep '^{' > $O/$name.json; echo "$name: $(python -c "import json,sys; d=json.load(open('$O/$name.json')); print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'build_s', d.get('build_s'))" 2>&1)"; }
run c2_cg_jacobi_256_csr --grid 256
run c2_cg_jacobi_256_ell --grid 256 --format ell
run c2_cg_jacobi_256_hyb --grid 256 --format hyb
run c4_bicgstab_mcsgs_512_ell --grid 512 --solver bicgstab --precond mcsgs --format ell --steps 60
run c4_bicgstab_mcsgs_512_csr --grid 512 --solver bicgstab --precond mcsgs --steps 60
run c3_gmres_ilu0_512 --grid 512 --solver gmres --precond ilu0 --steps 60
run gmres_mcilu_512 --grid 512 --solver gmres --precond mcilu --steps 60
run bicgstab_mcgs_512 --grid 512 --solver bicgstab --precond mcgs --steps 60
run c5_mixed_512 --grid 512 --solver mixed --steps 10 --warmup 2
run cg_jacobi_512_ell --grid 512 --format ell
run cg_jacobi_512_hyb --grid 512 --format hyb
run cg_ic_256 --grid 256 --precond ic --steps 60
run cg_ic_512 --grid 512 --precond ic --steps 60
run cg_jacobi_512_dia --grid 512 --format dia
run cg_jacobi_256_dia --grid 256 --format dia
