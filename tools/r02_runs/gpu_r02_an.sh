#!/bin/bash
mkdir -p gpurun_out/r02an
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_box_tiles_forced.py tests/test_gpu_shell.py -x -q -m gpu > gpurun_out/r02an/forced.log 2>&1; echo "forced+shell tests rc=$?"; tail -3 gpurun_out/r02an/forced.log
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --solver gmres --precond ilu0 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02an/b_p.json 2> gpurun_out/r02an/b_p.err; echo "bench poisson rc=$?"; grep "box-tile plan (" gpurun_out/r02an/b_p.err | tail -1 | cut -c1-200; python -c "import sys,json; d=json.loads(open('gpurun_out/r02an/b_p.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['final_residual'], 'build_s', d['build_s'])"
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 30 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02an/b_s.json 2> gpurun_out/r02an/b_s.err; echo "bench shell rc=$?"; grep "box-tile plan (" gpurun_out/r02an/b_s.err | tail -1 | cut -c1-200; python -c "import sys,json; d=json.loads(open('gpurun_out/r02an/b_s.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['final_residual'], 'build_s', d['build_s'])"
