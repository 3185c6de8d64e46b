# direct (level-scheduled) against iterative (Jacobi-sweep) triangular solves inside GMRES+ILU(0) / CG+IC, Poisson 512^3
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/it
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras "$@" >gpurun_out/it/$name.out 2>gpurun_out/it/$name.err; echo "$name rc=$?"; grep '^{' gpurun_out/it/$name.out > gpurun_out/it/$name.json; python -c "import json; d=json.load(open('gpurun_out/it/$name.json')); print('$name', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'build_s', d.get('build_s'))"; }
run gmres_ilu0_direct --grid 512 --solver gmres --precond ilu0 --steps 60
run gmres_ilu0_it3 --grid 512 --solver gmres --precond ilu0 --steps 60 --itsolve 3
run gmres_ilu0_it5 --grid 512 --solver gmres --precond ilu0 --steps 60 --itsolve 5
run gmres_ilu0_it10 --grid 512 --solver gmres --precond ilu0 --steps 60 --itsolve 10
run cg_ic_it5 --grid 512 --precond ic --steps 60 --itsolve 5
run cg_sgs_it5 --grid 512 --precond sgs --steps 60 --itsolve 5
run cg_ic_direct --grid 512 --precond ic --steps 60
run gmres_ilu0_direct2 --grid 512 --solver gmres --precond ilu0 --steps 60
