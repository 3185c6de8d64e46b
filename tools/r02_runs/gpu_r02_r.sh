#!/bin/bash
mkdir -p gpurun_out/r02r
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 4 --warmup 2 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02r/prof.json 2> gpurun_out/r02r/prof.err; echo "prof rc=$?"; grep "trsv prof" gpurun_out/r02r/prof.err | tail -4
RAMD_TRSV_NOFILL=1 RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 4 --warmup 2 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02r/prof_nf.json 2> gpurun_out/r02r/prof_nf.err; echo "prof nofill rc=$?"; grep "trsv prof" gpurun_out/r02r/prof_nf.err | tail -4
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 1500 python -m pytest tests -x -q -m gpu -k "lusolve or lsolve or usolve or ilu or ic or sgs or tri or precond or shell" > gpurun_out/r02r/forced_ct.log 2>&1; echo "forced ct rc=$?"; tail -5 gpurun_out/r02r/forced_ct.log
