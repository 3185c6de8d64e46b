#!/bin/bash
mkdir -p gpurun_out/r02p
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02p/forced_ct.log 2>&1; echo "forced ct rc=$?"; tail -15 gpurun_out/r02p/forced_ct.log
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02p/bench_gmres.json 2> gpurun_out/r02p/bench_gmres.err; echo "bench gmres rc=$?"; tail -3 gpurun_out/r02p/bench_gmres.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02p/bench_gmres.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'])"
RAMD_TRSV_NOFILL=1 timeout 900 python bench.py --solver gmres --precond ilu0 --steps 20 --warmup 5 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02p/bench_gmres_nofill.json 2> gpurun_out/r02p/bench_gmres_nofill.err; echo "nofill rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02p/bench_gmres_nofill.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'])"
