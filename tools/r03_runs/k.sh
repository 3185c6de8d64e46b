#!/bin/bash
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/placement_probe.py 14 2>&1 | grep -v "^addresses\|^offsets" | tee $O/probe1.log
timeout 300 python tools/placement_probe.py 14 2>&1 | grep "placement classes\|SAME" | tee $O/probe2.log
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v=d['kernels']['vector_updates']
print(sys.argv[2], 'cg', d['value'], 'ms/it', d['ms_per_step'], 'spmv', d['roofline']['avg_ms'], d['roofline']['frac'], 'vec', v['avg_ms'], '| cols-read', d.get('columns_read'), (d.get('roofline_columns_read') or {}).get('avg_ms'))
PY
}
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/a$i.json 2> $O/a$i.err; line $O/a$i.json "classes run$i"
done
for i in 1 2; do
  RAMD_ALLOC_CLASSES=0 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/n$i.json 2> $O/n$i.err; line $O/n$i.json "noclasses run$i"
done
timeout 1500 python tools/r03_runs/i.py > $O/dist_debug.log 2>&1; grep "max diff\|FAILED" $O/dist_debug.log
