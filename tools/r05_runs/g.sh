cd /root/repo
timeout 900 python -m pytest tests/test_gpu_solvers.py -x -q -m gpu -k "pairs_forced" 2>&1 | tail -3
RAMD_CSR_PAT=1 timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_kernels.py -x -q -m gpu -k "(mcsgs or mcgs or mcilu or multicolor) and not forced and not fresh_process" 2>&1 | tail -3
for f in 1 0 1 0; do
RAMD_MC_FOLD=$f timeout 600 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fold=$f', d['value'],'it/s', 'apply', d['kernels']['precond_apply']['avg_ms'], 'spmv', d['roofline']['avg_ms'], 'res', d['final_residual'])"
done
