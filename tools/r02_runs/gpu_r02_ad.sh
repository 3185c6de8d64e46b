#!/bin/bash
mkdir -p gpurun_out/r02ad
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ad/bench_shell.json 2> gpurun_out/r02ad/bench_shell.err; echo "bench shell rc=$?"; grep "box-tile" gpurun_out/r02ad/bench_shell.err | tail -3; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ad/bench_shell.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['build_s'])"
RAMD_TRSV_NOFILL=1 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ad/bench_shell_nf.json 2> gpurun_out/r02ad/bench_shell_nf.err; echo "bench shell nofill rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ad/bench_shell_nf.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'])"
