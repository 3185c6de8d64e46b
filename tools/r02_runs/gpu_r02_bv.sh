#!/bin/bash
# bench.py with the warm-up inside the timed Solve (time mark at iteration W): the driver's command line, the default, the
# Global path at one rank, GMRES and the mixed-precision line
mkdir -p gpurun_out/r02bv
cd /root/repo
export TMPDIR=/tmp
show() { python -c "import sys,json; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['steps'], d['warmup'], d['roofline']['avg_ms'], d.get('final_residual'))"; }
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02bv/drv.json 2> gpurun_out/r02bv/drv.err; echo rc=$?; show gpurun_out/r02bv/drv.json driver
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 0 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bv/w0.json 2> gpurun_out/r02bv/w0.err; echo rc=$?; show gpurun_out/r02bv/w0.json warmup0
timeout 900 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bv/def.json 2> gpurun_out/r02bv/def.err; echo rc=$?; show gpurun_out/r02bv/def.json default
timeout 900 python bench.py --force-global --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bv/glob.json 2> gpurun_out/r02bv/glob.err; echo rc=$?; show gpurun_out/r02bv/glob.json global1
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bv/gm.json 2> gpurun_out/r02bv/gm.err; echo rc=$?; show gpurun_out/r02bv/gm.json gmres
timeout 900 python bench.py --solver mixed --steps 6 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras > gpurun_out/r02bv/mx.json 2> gpurun_out/r02bv/mx.err; echo rc=$?; show gpurun_out/r02bv/mx.json mixed
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02bv/sh.json 2> gpurun_out/r02bv/sh.err; echo rc=$?; show gpurun_out/r02bv/sh.json shell
