cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/it
timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras --grid 512 --solver gmres --precond ilu0 --steps 60 2>&1 | tail -5 | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras --grid 512 --solver gmres --precond ilu0 --steps 60 --itsolve 5 2>&1 | tail -5 | cut -c1-400
