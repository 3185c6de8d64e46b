#!/bin/bash
# blocked MGS (ramd_fused_mgs_block): unit tests, the GMRES parity tests, A/B of the bench lines on one box
mkdir -p gpurun_out/r02bk
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "mgs" > gpurun_out/r02bk/t1.log 2>&1; echo "mgs tests rc=$?"; tail -3 gpurun_out/r02bk/t1.log
timeout 2400 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_shell.py tests/test_gpu_distributed.py -x -q -m gpu -k "gmres or GMRES or shell or fgmres" > gpurun_out/r02bk/t2.log 2>&1; echo "gmres tests rc=$?"; tail -3 gpurun_out/r02bk/t2.log
for blk in 0 1; do
RAMD_MGS_BLOCK=$blk timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bk/b_$blk.json 2> gpurun_out/r02bk/b_$blk.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bk/b_$blk.json').read().strip().splitlines()[-1]); print('block=$blk 512', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['final_residual'])"
RAMD_MGS_BLOCK=$blk timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bk/s_$blk.json 2> gpurun_out/r02bk/s_$blk.err; python -c "import sys,json; d=json.loads(open('gpurun_out/r02bk/s_$blk.json').read().strip().splitlines()[-1]); print('block=$blk shell', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['final_residual'])"
done
