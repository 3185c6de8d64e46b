#!/bin/bash
mkdir -p gpurun_out/r02w
cd /root/repo
export TMPDIR=/tmp
touch rocalution_amd/csrc/trisolve.hip
RAMD_EXTRA_CXXFLAGS="-DRAMD_CT_X=1" python -m rocalution_amd.build > gpurun_out/r02w/rebuild.log 2>&1; echo "rebuild rc=$?"
tag=x1
RAMD_TRSV_NOFILL=1 RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02w/nf_$tag.json 2> gpurun_out/r02w/nf_$tag.err; echo "$tag prof nofill rc=$?"; grep "trsv prof (" gpurun_out/r02w/nf_$tag.err | tail -2
RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02w/pf_$tag.json 2> gpurun_out/r02w/pf_$tag.err; echo "$tag prof rc=$?"; grep "trsv prof (" gpurun_out/r02w/pf_$tag.err | tail -2
