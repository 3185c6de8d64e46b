cd ${GRAFT_REPO_ROOT:-.}
run() { echo -n "$1: "; env $1 timeout 300 python bench.py --no-cpu-baseline --no-reference-gpu --no-extras --steps 100 --warmup 10 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['avg_ms'])"; }
run RAMD_EW_GRID_MULT=16
run RAMD_EW_GRID_MULT=8
run RAMD_EW_GRID_MULT=32
run RAMD_EW_GRID_MULT=64
run RAMD_EW_GRID_MULT=4
run RAMD_NT_STORES=1
run RAMD_EW_GRID_MULT=16
