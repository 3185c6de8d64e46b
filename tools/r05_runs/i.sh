cd /root/repo
RAMD_TRSV_BAND=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 RAMD_TRSV_CT_VERBOSE=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_shell.py -x -q -s -m gpu -k "(ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0 or variants) and not full_size and not cpp and not fresh_process and not forced" 2>&1 | grep -v "^band plan\|^box-tile\|^lattice" | tail -8
RAMD_TRSV_BAND=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 RAMD_TRSV_CT_VERBOSE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -s -m gpu -k "lusolve" 2>&1 | grep -c "^band plan"
mkdir -p gpurun_out/shellv
for v in rcm delaunay lex random; do
  RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --shell-variant $v --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu --no-cpu-baseline > gpurun_out/shellv/$v.json 2> gpurun_out/shellv/$v.err
  echo "$v rc=$?"; grep "plan (" gpurun_out/shellv/$v.err | cut -c1-200; grep '^{' gpurun_out/shellv/$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'it/s', d['ms_per_step'],'ms build',d['build_s'], 'trsv avg', d['roofline']['avg_ms'], d['tri_plan']['lower']['form'], 'res', d['final_residual'])"
done
