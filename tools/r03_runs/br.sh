#!/bin/bash
O=gpurun_out/r03br; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "variants_forced and XL" > $O/t1.log 2>&1; echo "xl variant tests rc=$?"; tail -3 $O/t1.log
for rep in 1 2 3 4; do for P in 0 1; do for D in 0 1; do
 RAMD_CSR_XL=$P DOT=$D TAG=xl=$P timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1 | sed 's/ (min.*algorithmic = / /; s/| norm.*| /| /'
done; done; done
