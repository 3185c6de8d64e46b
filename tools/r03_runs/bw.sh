#!/bin/bash
export TMPDIR=/tmp
for v in "RAMD_TRSV_WSLOT=0 RAMD_TRSV_NOFILL=1" "RAMD_TRSV_WSLOT=1 RAMD_TRSV_NOFILL=1" "RAMD_TRSV_WSLOT=0 RAMD_TRSV_NOFILL=1" "RAMD_TRSV_WSLOT=1 RAMD_TRSV_NOFILL=1"; do
env TAG="$v" $v timeout 600 python tools/trsv_time.py poisson 512 2>&1 | tail -1 | cut -d'|' -f2-
done
for v in "RAMD_TRSV_WSLOT=0 RAMD_TRSV_PROF=1" "RAMD_TRSV_WSLOT=1 RAMD_TRSV_PROF=1"; do
env TAG="$v" $v timeout 600 python tools/trsv_time.py poisson 512 2>&1 | grep "trsv prof (" | tail -2
done
