#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
export TMPDIR=/tmp
timeout 3400 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/gpu_suite.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; tail -c 3000 $O/bench_default.json
