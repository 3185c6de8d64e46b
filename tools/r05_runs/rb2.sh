R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rb; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lattice.py -m gpu -x -q -k red_black 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_full_size.py tests/test_gpu_global.py -m gpu -x -q 2>&1 | tail -5
RAMD_SLAB_ONLY=bicgstab timeout 300 python tools/slab_probe.py
RAMD_MC_RB=0 RAMD_SLAB_ONLY=bicgstab timeout 300 python tools/slab_probe.py
cd /tmp
for f in csr ell hyb; do
  timeout 600 python $R/bench.py --format $f --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu 2> $O/f_$f.err | grep '^{' > $O/f_$f.json
  python3 -c "
import json; d=json.load(open('$O/f_$f.json')); print('$f', d['value'], 'it/s', d['ms_per_step'], 'ms', {k:(v.get('avg_ms') if isinstance(v,dict) else v) for k,v in d.get('kernels',{}).items()}, d['roofline'])"
done
