#!/bin/bash
mkdir -p gpurun_out/r02bf
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py tests/test_gpu_kernels.py -x -q -m gpu -k "forced or ragged or fp32_and or shell or 27_point or lusolve or trisolve" > gpurun_out/r02bf/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02bf/t.log
for i in 1 2; do timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bf/s.json 2> gpurun_out/r02bf/s.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bf/s.json').read().strip().splitlines()[-1]); print('shell', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'], d['final_residual'])"; done
