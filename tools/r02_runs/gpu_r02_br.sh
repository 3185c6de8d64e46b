#!/bin/bash
# what the driver runs at round end: default bench.py (wall time), smoke; plus the RAMD_MGS_BLOCK=0 solver tests
mkdir -p gpurun_out/r02br
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_solvers.py -x -q -m gpu -k "one_projection" > gpurun_out/r02br/t.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r02br/t.log
s=$(date +%s); timeout 1200 python bench.py > gpurun_out/r02br/bench.json 2> gpurun_out/r02br/bench.err; echo "default bench rc=$? wall=$(( $(date +%s) - s )) s"
python -c "
import json; d=json.loads(open('gpurun_out/r02br/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'], d['cpu_baseline'], d.get('extras'), d.get('reference_gpu'))"
