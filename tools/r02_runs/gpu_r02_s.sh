#!/bin/bash
mkdir -p gpurun_out/r02s
cd /root/repo
RAMD_TRSV_NOFILL=1 RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02s/prof_nf.json 2> gpurun_out/r02s/prof_nf.err; echo "prof nofill rc=$?"; grep "trsv prof" gpurun_out/r02s/prof_nf.err | tail -4
