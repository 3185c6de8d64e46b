#!/bin/bash
mkdir -p gpurun_out/r02bc
cd /root/repo
export TMPDIR=/tmp
run() {
  tag=$1
  timeout 900 python bench.py --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bc/b_$tag.json 2> gpurun_out/r02bc/b_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bc/b_$tag.json').read().strip().splitlines()[-1]); print('$tag 512', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
  timeout 900 python bench.py --solver gmres --precond ilu0 --grid 256 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bc/c_$tag.json 2> gpurun_out/r02bc/c_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bc/c_$tag.json').read().strip().splitlines()[-1]); print('$tag 256', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
  timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 40 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bc/s_$tag.json 2> gpurun_out/r02bc/s_$tag.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bc/s_$tag.json').read().strip().splitlines()[-1]); print('$tag shell', d['value'], d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
}
run cap64
for c in 8 2; do
touch rocalution_amd/csrc/trisolve.hip
RAMD_EXTRA_CXXFLAGS="-DRAMD_CT_POLL_CAP=$c" python -m rocalution_amd.build > gpurun_out/r02bc/rebuild.log 2>&1; echo "rebuild rc=$?"
run cap$c
done
