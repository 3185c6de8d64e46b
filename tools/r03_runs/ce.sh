#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do
for v in "RAMD_CSR_W4=0" "RAMD_CSR_W4=1 RAMD_CSR_W4_WAVES=4" "RAMD_CSR_W4=1 RAMD_CSR_W4_WAVES=1"; do
env TAG="$v" $v timeout 300 python tools/spmv_shell.py 549 2>&1 | tail -2 | cut -c1-60,95-
done; done
