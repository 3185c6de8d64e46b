#!/bin/bash
mkdir -p gpurun_out/r02ay
cd /root/repo
export TMPDIR=/tmp
run() {
  tag=$1
  for i in 1 2; do
  timeout 900 python bench.py --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02ay/b_$tag.json 2> gpurun_out/r02ay/b_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02ay/b_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
  done
}
run base
for v in "-DRAMD_CT_DEPTH3=12" "-DRAMD_CT_DEPTH3=6" "-DRAMD_CT_RING=4" "-DRAMD_CT_RING=2"; do
  touch rocalution_amd/csrc/trisolve.hip
  RAMD_EXTRA_CXXFLAGS="$v" python -m rocalution_amd.build > gpurun_out/r02ay/rebuild.log 2>&1 || { echo "build failed $v"; tail -5 gpurun_out/r02ay/rebuild.log; continue; }
  run "$v"
done
