#!/bin/bash
# blocked MGS with blocks of 8: unit + GMRES tests, bench lines; then the same bench with the library rebuilt at K = 4
mkdir -p gpurun_out/r02bn
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mgs" > gpurun_out/r02bn/t1.log 2>&1; echo "mgs tests rc=$?"; tail -3 gpurun_out/r02bn/t1.log
timeout 2400 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_shell.py tests/test_gpu_distributed.py -x -q -m gpu -k "gmres or GMRES or shell or fgmres" > gpurun_out/r02bn/t2.log 2>&1; echo "gmres tests rc=$?"; tail -3 gpurun_out/r02bn/t2.log
run() {
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bn/b_$1.json 2> gpurun_out/r02bn/b_$1.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bn/b_$1.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; print('$1 512', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], v['avg_ms'], v['achieved'], d['final_residual'])"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bn/s_$1.json 2> gpurun_out/r02bn/s_$1.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bn/s_$1.json').read().strip().splitlines()[-1]); print('$1 shell', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['final_residual'])"
}
run k8
RAMD_EXTRA_CXXFLAGS="-DRAMD_MGS_K=4" python -m rocalution_amd.build --force > gpurun_out/r02bn/rebuild.log 2>&1; echo "rebuild rc=$?"
run k4
