#!/bin/bash
O=gpurun_out/r03w; mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v=d['kernels']['vector_updates']
print(sys.argv[2], 'cg', d['value'], 'ms/it', d['ms_per_step'], 'spmv', d['roofline']['avg_ms'], d['roofline']['frac'], 'vec', v['avg_ms'], '| cols-read', d.get('columns_read'), (d.get('roofline_columns_read') or {}).get('avg_ms'))
PY
}
for i in 1 2 3 4 5 6 7 8 9 10; do
  RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/a$i.json 2> $O/a$i.err; line $O/a$i.json "run$i"
  grep -h "place" $O/a$i.err | head -6
done
timeout 900 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/bicgstab.json 2> $O/bicgstab.err; echo "bicgstab rc=$?"
timeout 900 python bench.py --force-global --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/global1.json 2> $O/global1.err; echo "global1 rc=$?"
python - <<'PY'
import json
for n in ('bicgstab','global1'):
    try:
        d=json.loads(open('gpurun_out/r03w/%s.json'%n).read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d.get('roofline',{}).get('avg_ms'), d.get('roofline',{}).get('frac'), {k:(v['avg_ms'],v['frac']) for k,v in d.get('kernels',{}).items()})
    except Exception as e: print(n, 'ERR', e)
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/gpu_suite.log
