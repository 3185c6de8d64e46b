cd /root/repo; mkdir -p gpurun_out/shellv
for v in rcm; do
  RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --shell-variant $v --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu --no-cpu-baseline > gpurun_out/shellv/$v.json 2> gpurun_out/shellv/$v.err
  echo "$v rc=$?"; grep "plan (" gpurun_out/shellv/$v.err | cut -c1-200; grep '^{' gpurun_out/shellv/$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'],'it/s', d['ms_per_step'],'ms build',d['build_s'], 'trsv avg', d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])"
done
