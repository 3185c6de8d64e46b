#!/bin/bash
O=gpurun_out/r03bh; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do for P in 1 3 4; do for D in 0 1; do
 RAMD_CSR_PAT2=$P DOT=$D TAG=nb=$P timeout 300 python tools/spmv_time.py 512 100 2>&1 | tail -1 | sed 's/ (min.*algorithmic = / /; s/| norm.*| /| /'
done; done; done
