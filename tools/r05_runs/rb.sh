# fused red-black MC-SGS: parity, then A/B bench
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/rb; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lattice.py -m gpu -x -q -k red_black 2>&1 | tail -15
RAMD_MC_RB=2 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -m gpu -x -q -k "mcsgs or MultiColored or multicolo or sgs" 2>&1 | tail -8
cd /tmp
for rb in 0 1; do
  RAMD_MC_RB=$rb timeout 600 python $R/bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu 2> $O/rb$rb.err | grep '^{' > $O/rb$rb.json
  python3 -c "
import json; d=json.load(open('$O/rb$rb.json')); print('RB=$rb', d['value'], 'it/s', d['ms_per_step'], 'ms', {k:(v.get('avg_ms') if isinstance(v,dict) else v) for k,v in d.get('kernels',{}).items()})"
done
