#!/bin/bash
# cost of the pattern analysis after the one-lane flag check: build times of MC-SGS and the first SpMV, pattern tests
mkdir -p gpurun_out/r02cb
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "row_patterns" > gpurun_out/r02cb/t1.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r02cb/t1.log
for pat in -1 0; do
RAMD_CSR_PAT=$pat timeout 900 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02cb/b_$pat.json 2> gpurun_out/r02cb/b_$pat.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02cb/b_$pat.json').read().strip().splitlines()[-1]); print('pat=$pat bicgstab+mcsgs', d['value'], d['ms_per_step'], 'build', d['build_s'])"
done
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02cb/kt -o b -- python /root/repo/bench.py --solver bicgstab --precond mcsgs --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras > /dev/null 2>&1; echo prof rc=$?
