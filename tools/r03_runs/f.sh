#!/bin/bash
# r03f: arenas (placement by construction): CG bench in fresh processes with and without; then the whole GPU suite
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v=d['kernels']['vector_updates']
print(sys.argv[2], 'cg', d['value'], 'ms/it', d['ms_per_step'], 'spmv', d['roofline']['avg_ms'], d['roofline']['frac'], 'vec', v['avg_ms'], '| cols-read', d.get('columns_read'), (d.get('roofline_columns_read') or {}).get('avg_ms'))
PY
}
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/a$i.json 2> $O/a$i.err; line $O/a$i.json "arena run$i"
done
for i in 1 2 3; do
  RAMD_ALLOC_ARENA=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/n$i.json 2> $O/n$i.err; line $O/n$i.json "noarena run$i"
done
for i in 1 2; do
  RAMD_CSR_XL=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/x$i.json 2> $O/x$i.err; line $O/x$i.json "arena xl=0 run$i"
done
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/gpu_suite.log
