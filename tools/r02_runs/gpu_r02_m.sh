#!/bin/bash
mkdir -p gpurun_out/r02m
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02m/forced_ct.log 2>&1; echo "forced ct rc=$?"; tail -3 gpurun_out/r02m/forced_ct.log
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02m/bench_gmres.json 2> gpurun_out/r02m/bench_gmres.err; echo "bench gmres rc=$?"; cat gpurun_out/r02m/bench_gmres.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels'])"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02m/bench_shell.json 2> gpurun_out/r02m/bench_shell.err; echo "bench shell rc=$?"; cat gpurun_out/r02m/bench_shell.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels'])"
RAMD_TRSV_NOFILL=1 timeout 900 python bench.py --solver gmres --precond ilu0 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02m/bench_gmres_nofill.json 2> gpurun_out/r02m/bench_gmres_nofill.err; echo "nofill rc=$?"; cat gpurun_out/r02m/bench_gmres_nofill.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernels'])"
