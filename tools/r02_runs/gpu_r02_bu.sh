#!/bin/bash
# kernel trace of BiCGStab + MC-SGS at 512^3 (config 4's solver on one GPU)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r02bu
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02bu/kt -o bench -- python $R/bench.py --no-cpu-baseline --no-reference-gpu --no-extras --solver bicgstab --precond mcsgs --steps 60 --warmup 10 > $R/gpurun_out/r02bu/bench.json 2> $R/gpurun_out/r02bu/err.log
echo rc=$?
