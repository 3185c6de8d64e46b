cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras "$@" 2>gpurun_out/$name.err | grep '^{' > gpurun_out/$name.json; python -c "import json; d=json.load(open('gpurun_out/$name.json')); print('$name', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'build_s', d.get('build_s'))"; }
run cg_ic_256 --grid 256 --precond ic --steps 60
run cg_ic_512 --grid 512 --precond ic --steps 60
run cg_sgs_512 --grid 512 --precond sgs --steps 60
run cg_ilu0_512 --grid 512 --precond ilu0 --steps 60
