#!/bin/bash
export TMPDIR=/tmp
for v in "8,8,8" "16,8,4" "16,4,8" "32,4,4" "16,8,8" "16,16,4" "8,8,8" "16,8,4"; do
env TAG="box=$v" RAMD_TRSV_CT_BOX=$v timeout 600 python tools/trsv_time.py poisson 512 2>&1 | tail -1 | cut -d'|' -f2-
done
