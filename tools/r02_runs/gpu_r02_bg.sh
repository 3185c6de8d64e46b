#!/bin/bash
mkdir -p gpurun_out/r02bg
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py -x -q -m gpu > gpurun_out/r02bg/t.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02bg/t.log
tag=$1
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bg/b_$tag.json 2> gpurun_out/r02bg/b_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bg/b_$tag.json').read().strip().splitlines()[-1]); print('$tag 512', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'], d['final_residual'])"
timeout 900 python bench.py --solver gmres --precond ilu0 --grid 256 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bg/c_$tag.json 2> gpurun_out/r02bg/c_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bg/c_$tag.json').read().strip().splitlines()[-1]); print('$tag 256', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bg/s_$tag.json 2> gpurun_out/r02bg/s_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bg/s_$tag.json').read().strip().splitlines()[-1]); print('$tag shell', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'], d['final_residual'])"
