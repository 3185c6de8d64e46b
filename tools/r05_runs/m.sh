cd /root/repo; mkdir -p gpurun_out/shellv
RAMD_TRSV_BAND=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_shell.py -x -q -m gpu -k "(ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0 or variants) and not full_size and not cpp and not fresh_process and not forced" 2>&1 | tail -3
for v in rcm delaunay; do
  timeout 900 python bench.py --matrix shell --shell-variant $v --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu --no-cpu-baseline > gpurun_out/shellv/$v.json 2> gpurun_out/shellv/$v.err
  grep '^{' gpurun_out/shellv/$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', d['value'],'it/s', d['ms_per_step'],'ms build',d['build_s'], 'trsv avg', d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'], 'res', d['final_residual'])"
done
