#!/bin/bash
# blocked MGS: block size 8 / 6 / 4 (library rebuilt on the box), sums accumulated with fma
mkdir -p gpurun_out/r02bo
cd /root/repo
export TMPDIR=/tmp
run() {
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bo/b_$1.json 2> gpurun_out/r02bo/b_$1.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bo/b_$1.json').read().strip().splitlines()[-1]); v=d['kernels']['vector_updates']; print('$1 512', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], v['avg_ms'], v['achieved'], d['final_residual'])"
}
run k8
for k in 6 4 5; do
RAMD_EXTRA_CXXFLAGS="-DRAMD_MGS_K=$k" python -m rocalution_amd.build --force > gpurun_out/r02bo/rebuild_$k.log 2>&1; echo "rebuild rc=$?"
run k$k
done
