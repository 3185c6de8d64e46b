#!/bin/bash
mkdir -p gpurun_out/r02ak
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 300 python tools/dbg_small.py 6 12 24 64 > gpurun_out/r02ak/small.log 2>&1; echo "small rc=$?"; tail -2 gpurun_out/r02ak/small.log
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shell.py -x -q -m gpu -k "lusolve or ilu or shell" > gpurun_out/r02ak/kern.log 2>&1; echo "kernels rc=$?"; tail -3 gpurun_out/r02ak/kern.log
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ak/b_p.json 2> gpurun_out/r02ak/b_p.err; echo "bench poisson rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ak/b_p.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'], d['build_s'])"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ak/b_s.json 2> gpurun_out/r02ak/b_s.err; echo "bench shell rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ak/b_s.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'], d['build_s'])"
