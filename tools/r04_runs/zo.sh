# round 4, call 45: cubic tile boxes by default -- the slab, the cube, and the triangular-solve tests
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zo
mkdir -p $O
cd $R
RAMD_SLAB_ONLY=gmres timeout 300 python tools/slab_probe.py 64 2>&1 | grep slab
RAMD_SLAB_ONLY=gmres timeout 300 python tools/slab_probe.py 128 2>&1 | grep slab
RAMD_SLAB_ONLY=gmres timeout 300 python tools/slab_probe.py 256 2>&1 | grep slab
timeout 600 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cube it/s', d['value'])"
timeout 2400 python -m pytest tests/test_gpu_box_tiles_forced.py tests/test_gpu_full_size.py tests/test_gpu_shell.py -m gpu -x -q > $O/t.log 2>&1; tail -3 $O/t.log
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -m gpu -x -q -k "lu or LU or ilu or trsv or tri or ic or IC" > $O/t2.log 2>&1; tail -3 $O/t2.log
