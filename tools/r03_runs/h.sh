#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
for v in "RAMD_CSR_XL=1" "RAMD_CSR_XL=0"; do for d in 0 1; do
  env $v DOT=$d TAG="$v" timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1
done; done
timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q -m gpu > $O/dist.log 2>&1; echo "dist rc=$?"; tail -4 $O/dist.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/gpu_suite.log
