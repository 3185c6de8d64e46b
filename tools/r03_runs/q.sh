#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
export TMPDIR=/tmp
TAG=grp timeout 300 python tools/spmv_shell.py 549 2>&1 | tail -2
TAG=nogrp RAMD_CSR_GRP=0 timeout 300 python tools/spmv_shell.py 549 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmv_variants_forced" > $O/variants.log 2>&1; echo "variants rc=$?"; tail -3 $O/variants.log
RAMD_CSR_GRP=1 timeout 900 python -m pytest tests/test_gpu_shell.py tests/test_gpu_kernels.py -x -q -m gpu -k "spmv or shell or csr_long" > $O/grp.log 2>&1; echo "grp rc=$?"; tail -3 $O/grp.log
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/b_shell.json 2> $O/b_shell.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03q/b_shell.json').read().strip().splitlines()[-1])
print('shell gmres', d['value'], d['ms_per_step'], 'trsv', d['roofline']['avg_ms'], 'spmv', d['kernels']['spmv']['avg_ms'], d['kernels']['spmv']['frac'], 'cols-read', d.get('columns_read'), (d.get('roofline_columns_read') or {}).get('avg_ms'))
PY
