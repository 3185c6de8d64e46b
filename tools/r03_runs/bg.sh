#!/bin/bash
O=gpurun_out/r03bg; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do for P in 0 1; do
 RAMD_CSR_PAT=0 RAMD_CSR_BLKRP=$P DOT=0 TAG=cols,blkrp=$P timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1 | sed 's/ (min.*algorithmic = / /; s/| norm.*| /| /'
 RAMD_CSR_PAT=0 RAMD_CSR_BLKRP=$P DOT=1 TAG=cols,blkrp=$P timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1 | sed 's/ (min.*algorithmic = / /; s/| norm.*| /| /'
done; done
for rep in 1 2; do for P in 0 1; do
 RAMD_CSR_BLKRP=$P TAG=blkrp=$P timeout 300 python tools/spmv_shell.py 2>&1 | tail -2 | sed 's/| state.*| /| /'
done; done
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_shell.py -x -q -m gpu > $O/t.log 2>&1; echo "tests rc=$?"; tail -3 $O/t.log
