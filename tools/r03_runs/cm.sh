#!/bin/bash
export TMPDIR=/tmp
for v in 1 0; do
echo "RAMD_SUM_PARTIALS=$v"
RAMD_SUM_PARTIALS=$v timeout 900 python tools/dbg_c4_ell.py 2>&1 | grep "^fmt"
done
