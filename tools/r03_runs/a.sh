#!/bin/bash
# r03a: grouped (supernode) box-tile triangular solve -- parity on the shell tests (forced box tiles), then the config-3 surrogate line
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shell.py -x -q -m gpu -k "not full" > $O/shell_small.log 2>&1; echo "shell small rc=$?"; tail -3 $O/shell_small.log
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_DEDUP=1 RAMD_TRSV_CT_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_shell.py -x -q -s -m gpu -k "not full" > $O/shell_forced.log 2>&1; echo "shell forced rc=$?"; tail -3 $O/shell_forced.log; grep -c "row groups" $O/shell_forced.log
timeout 900 python -m pytest tests/test_gpu_box_tiles_forced.py -x -q -m gpu -k "fp32_and_fp64" > $O/cross.log 2>&1; echo "cross rc=$?"; tail -3 $O/cross.log
timeout 1200 python -m pytest tests/test_gpu_shell.py -x -q -m gpu -k "full" > $O/shell_full.log 2>&1; echo "shell full rc=$?"; tail -3 $O/shell_full.log
RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/b_shell.json 2> $O/b_shell.err; echo "bench rc=$?"
grep "box-tile" $O/b_shell.err | head
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03a/b_shell.json').read().strip().splitlines()[-1])
print('shell gmres', d['value'], d['ms_per_step'], d.get('roofline'))
PY
RAMD_TRSV_CT_GROUPS=0 timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > $O/b_shell_nogrp.json 2> $O/b_shell_nogrp.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03a/b_shell_nogrp.json').read().strip().splitlines()[-1])
print('shell gmres (no groups)', d['value'], d['ms_per_step'], d.get('roofline'))
PY
