#!/bin/bash
mkdir -p gpurun_out/r02bi
cd /root/repo
export TMPDIR=/tmp
run() {
  tag=$1
  timeout 900 python bench.py --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bi/b_$tag.json 2> gpurun_out/r02bi/b_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bi/b_$tag.json').read().strip().splitlines()[-1]); print('$tag 512', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
  timeout 900 python bench.py --solver gmres --precond ilu0 --grid 256 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bi/c_$tag.json 2> gpurun_out/r02bi/c_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bi/c_$tag.json').read().strip().splitlines()[-1]); print('$tag 256', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
  timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bi/s_$tag.json 2> gpurun_out/r02bi/s_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bi/s_$tag.json').read().strip().splitlines()[-1]); print('$tag shell', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
}
for v in "4 4" "3 3" "5 5"; do
  set -- $v
  touch rocalution_amd/csrc/trisolve.hip
  RAMD_EXTRA_CXXFLAGS="-DRAMD_CT_DEPTH3=$1 -DRAMD_CT_DEPTH8L=$2" python -m rocalution_amd.build > gpurun_out/r02bi/rebuild.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/r02bi/rebuild.log; continue; }
  run d$1
done
