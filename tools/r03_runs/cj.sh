#!/bin/bash
O=gpurun_out/r03cj; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2 3 4; do
for v in 0 1; do
RAMD_SUM_PARTIALS=$v timeout 600 python bench.py --steps 100 --warmup 10 $B > $O/cg_${v}_$i.json 2> $O/cg_${v}_$i.err
python - $O/cg_${v}_$i.json $v <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print('sum_partials=%s' % sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['avg_ms'], 'placement', d.get('placement_s'), 'iters', d.get('iterations'), 'res', d.get('final_residual'))
PY
done; done
for v in 0 1; do
RAMD_SUM_PARTIALS=$v timeout 600 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('bicgstab sum_partials=$v', d['value'], d['ms_per_step'])"
done
timeout 1500 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py tests/test_gpu_kernels.py -x -q -m gpu -k "not fresh_process" > $O/t1.log 2>&1; echo "tests rc=$?"; tail -3 $O/t1.log
