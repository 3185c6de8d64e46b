#!/bin/bash
mkdir -p gpurun_out/r02ai
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shell.py -x -q -m gpu > gpurun_out/r02ai/shell_tests.log 2>&1; echo "shell tests rc=$?"; tail -3 gpurun_out/r02ai/shell_tests.log
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ai/bench_shell.json 2> gpurun_out/r02ai/bench_shell.err; echo "bench shell rc=$?"; grep "box-tile plan" gpurun_out/r02ai/bench_shell.err | tail -4 | cut -c1-300; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ai/bench_shell.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'], d['build_s'])"
RAMD_TRSV_CT_DEDUP=1 RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02ai/forced_ct_dedup.log 2>&1; echo "forced ct + dedup rc=$?"; tail -3 gpurun_out/r02ai/forced_ct_dedup.log
