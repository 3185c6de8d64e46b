#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
v=d['kernels']['vector_updates']
print(sys.argv[2], 'cg', d['value'], 'ms/it', d['ms_per_step'], 'spmv', d['roofline']['avg_ms'], d['roofline']['frac'], 'vec', v['avg_ms'], '| cols-read', d.get('columns_read'), (d.get('roofline_columns_read') or {}).get('avg_ms'))
PY
}
for i in 1 2 3 4 5 6 7 8; do
  RAMD_ALLOC_VERBOSE=1 timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/a$i.json 2> $O/a$i.err; line $O/a$i.json "classes run$i"
  grep "class probe" $O/a$i.err | awk '{print $6, $7, $NF}' | tr '\n' ';' | cut -c1-400; echo
done
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/dist.log 2>&1; echo "dist rc=$?"; tail -4 $O/dist.log
