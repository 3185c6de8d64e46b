#!/bin/bash
# r03c: x tiles in LDS for the structured CSR product (k_csr_xl) + cross-section of the grouped triangular solve
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
for v in "RAMD_CSR_XL=1" "RAMD_CSR_XL=0" "RAMD_CSR_PAT=0"; do
  env $v TAG="$v" timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1
  env $v TAG="$v" timeout 300 python tools/spmv_time.py 256 200 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmv_variants_forced" > $O/variants.log 2>&1; echo "variants rc=$?"; tail -3 $O/variants.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q -m gpu -k "spmv or closed" > $O/full.log 2>&1; echo "full rc=$?"; tail -3 $O/full.log
run() { TAG="$1" timeout 300 python tools/trsv_time.py shell 549 2> $O/err_$2.log | tail -1; grep "box-tile plan (" $O/err_$2.log | sed 's/.*extents/extents/' | head -1; }
RAMD_TRSV_CT_VERBOSE=1 run "default(cross3)" 0
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GCROSS=2 run "cross2" 1
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GCROSS=6 run "cross6" 2
RAMD_TRSV_CT_VERBOSE=1 RAMD_TRSV_CT_GCROSS=3 RAMD_TRSV_CT_ROWS=1024 RAMD_TRSV_CT_LDS=65536 run "cross3 rows1024" 3
for v in "RAMD_CSR_XL=1" "RAMD_CSR_XL=0"; do
 env $v timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-reference-gpu --no-extras > $O/cg_$v.json 2> $O/cg_$v.err; python - <<PY
import json
d=json.loads(open('$O/cg_$v.json').read().strip().splitlines()[-1])
print('$v cg', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], d['roofline']['frac'], d.get('columns_read'), (d.get('roofline_columns_read') or {}).get('avg_ms'))
PY
done
