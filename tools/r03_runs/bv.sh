#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3 4 5; do
for v in "RAMD_TRSV_WSLOT=0" "RAMD_TRSV_WSLOT=1"; do
env TAG="$v" $v timeout 600 python tools/trsv_time.py poisson 512 2>&1 | tail -1 | cut -d'|' -f2-
done; done
