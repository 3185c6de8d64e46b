#!/bin/bash
mkdir -p gpurun_out/r02aa
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02aa/forced_ct.log 2>&1; echo "forced ct rc=$?"; tail -5 gpurun_out/r02aa/forced_ct.log
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02aa/bench_shell.json 2> gpurun_out/r02aa/bench_shell.err; echo "bench shell rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02aa/bench_shell.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'])"
