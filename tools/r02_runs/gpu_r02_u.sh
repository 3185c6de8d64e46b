#!/bin/bash
mkdir -p gpurun_out/r02u
cd /root/repo
export TMPDIR=/tmp
tag=streams
RAMD_TRSV_NOFILL=1 RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02u/nf_$tag.json 2> gpurun_out/r02u/nf_$tag.err; echo "$tag prof nofill rc=$?"; grep "trsv prof (" gpurun_out/r02u/nf_$tag.err | tail -2
RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02u/pf_$tag.json 2> gpurun_out/r02u/pf_$tag.err; echo "$tag prof rc=$?"; grep "trsv prof (" gpurun_out/r02u/pf_$tag.err | tail -2
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02u/b_$tag.json 2> gpurun_out/r02u/b_$tag.err; echo "$tag bench rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02u/b_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'])"
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02u/forced_ct.log 2>&1; echo "forced ct rc=$?"; tail -5 gpurun_out/r02u/forced_ct.log
