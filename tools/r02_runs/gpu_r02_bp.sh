#!/bin/bash
# reduction grid size (partials per launch) and loads in flight of k_mgs_block: rebuild on the box, GMRES and CG bench lines
mkdir -p gpurun_out/r02bp
cd /root/repo
export TMPDIR=/tmp
run() {
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bp/b_$1.json 2> gpurun_out/r02bp/b_$1.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bp/b_$1.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; print('$1 gmres', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], v['avg_ms'], v['achieved'])"
timeout 900 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bp/c_$1.json 2> gpurun_out/r02bp/c_$1.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bp/c_$1.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; print('$1 cg', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], v['avg_ms'], v['achieved'])"
}
run base
RAMD_EXTRA_CXXFLAGS="-DRAMD_REDUCE_BLOCKS=16384" python -m rocalution_amd.build --force > gpurun_out/r02bp/rebuild_1.log 2>&1; echo "rebuild rc=$?"
run rb16k
RAMD_EXTRA_CXXFLAGS="-DRAMD_REDUCE_BLOCKS=32768" python -m rocalution_amd.build --force > gpurun_out/r02bp/rebuild_2.log 2>&1; echo "rebuild rc=$?"
run rb32k
RAMD_EXTRA_CXXFLAGS="-DRAMD_MGS_U=4" python -m rocalution_amd.build --force > gpurun_out/r02bp/rebuild_3.log 2>&1; echo "rebuild rc=$?"
run u4
RAMD_EXTRA_CXXFLAGS="-DRAMD_MGS_U=1" python -m rocalution_amd.build --force > gpurun_out/r02bp/rebuild_4.log 2>&1; echo "rebuild rc=$?"
run u1
