#!/bin/bash
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
for i in 1 2; do
for v in 8192 4096 2048 1024; do
RAMD_FUSED_GRID_CAP=$v timeout 600 python bench.py --steps 100 --warmup 10 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('cap=$v', d['value'], d['ms_per_step'], d['roofline']['avg_ms'], 'placement', d.get('placement_s'))"
done; done
for v in 8192 2048; do
RAMD_FUSED_GRID_CAP=$v timeout 600 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('gmres cap=$v', d['value'], d['ms_per_step'])"
RAMD_FUSED_GRID_CAP=$v timeout 600 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('bicgstab cap=$v', d['value'], d['ms_per_step'])"
done
