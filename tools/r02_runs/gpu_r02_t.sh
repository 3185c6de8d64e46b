#!/bin/bash
mkdir -p gpurun_out/r02t
cd /root/repo
export TMPDIR=/tmp
run() {
  tag=$1
  RAMD_TRSV_NOFILL=1 RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02t/nf_$tag.json 2> gpurun_out/r02t/nf_$tag.err; echo "$tag prof nofill rc=$?"; grep "trsv prof (" gpurun_out/r02t/nf_$tag.err | tail -2
  RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 2 --warmup 1 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02t/pf_$tag.json 2> gpurun_out/r02t/pf_$tag.err; echo "$tag prof rc=$?"; grep "trsv prof (" gpurun_out/r02t/pf_$tag.err | tail -2
  timeout 900 python bench.py --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02t/b_$tag.json 2> gpurun_out/r02t/b_$tag.err; echo "$tag bench rc=$?"; python -c "import sys,json; d=json.loads(open('gpurun_out/r02t/b_$tag.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['final_residual'])"
}
run mask1
touch rocalution_amd/csrc/trisolve.hip
RAMD_EXTRA_CXXFLAGS="-DRAMD_CT_MASK=0" python -m rocalution_amd.build > gpurun_out/r02t/rebuild.log 2>&1; echo "rebuild rc=$?"
run mask0
