#!/bin/bash
export TMPDIR=/tmp
for f in csr ell hyb; do for D in 0 1; do DOT=$D TAG=$f timeout 300 python tools/spmv_time.py 512 100 $f 2>&1 | tail -1 | sed 's/ (min.*algorithmic = / /; s/| norm.*| /| /'; done; done
