#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python tools/r03_runs/i.py 2>&1 | grep -v "^$" | tail -30
for ph in 32 0 8 16 24 48 64 96 128 160; do
  RAMD_ARENA_PHASE_MB=$ph timeout 300 python tools/placement_probe.py 14 2>&1 | grep "cg_update min"
done
RAMD_ALLOC_ARENA=0 timeout 300 python tools/placement_probe.py 14 2>&1 | grep "cg_update min"
RAMD_ARENA_PHASE_MB=32 RAMD_ARENA_MOD_MB=1024 timeout 300 python tools/placement_probe.py 14 2>&1 | grep "cg_update min"
RAMD_ARENA_PHASE_MB=64 RAMD_ARENA_MOD_MB=1024 timeout 300 python tools/placement_probe.py 14 2>&1 | grep "cg_update min"
RAMD_ARENA_PHASE_MB=16 RAMD_ARENA_MOD_MB=256 timeout 300 python tools/placement_probe.py 14 2>&1 | grep "cg_update min"
