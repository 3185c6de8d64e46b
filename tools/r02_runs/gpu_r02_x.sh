#!/bin/bash
mkdir -p gpurun_out/r02x
cd /root/repo
export TMPDIR=/tmp
tag=${1:-x}
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 300 python tools/dbg_small.py 6 12 24 64 > gpurun_out/r02x/small_$tag.log 2>&1; echo "small rc=$?"; grep -v "box-tile" gpurun_out/r02x/small_$tag.log | tail -5
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lusolve or ilu" > gpurun_out/r02x/kern_$tag.log 2>&1; echo "kernels rc=$?"; tail -3 gpurun_out/r02x/kern_$tag.log
RAMD_TRSV_NOFILL=1 RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02x/nf_$tag.json 2> gpurun_out/r02x/nf_$tag.err; echo "$tag prof nofill rc=$?"; grep "trsv prof (" gpurun_out/r02x/nf_$tag.err | tail -2
RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02x/pf_$tag.json 2> gpurun_out/r02x/pf_$tag.err; echo "$tag prof rc=$?"; grep "trsv prof (" gpurun_out/r02x/pf_$tag.err | tail -2
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02x/b_$tag.json 2> gpurun_out/r02x/b_$tag.err; echo "$tag bench rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02x/b_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'], d['build_s'])"
