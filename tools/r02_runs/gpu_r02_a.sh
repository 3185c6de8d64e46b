#!/bin/bash
# round-2 first GPU pass: new config-3 tests, streaming microbench, bench harness on the three workloads
mkdir -p gpurun_out/r02a
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shell.py -x -q > gpurun_out/r02a/shell_tests.log 2>&1; echo "shell tests rc=$?"
timeout 300 tools/_bin/membench > gpurun_out/r02a/membench.txt 2>&1; echo "membench rc=$?"
timeout 900 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 > gpurun_out/r02a/bench_shell.json 2> gpurun_out/r02a/bench_shell.err; echo "bench shell rc=$?"
timeout 900 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-reference-gpu --cpu-grid 256 --cpu-iters 20 > gpurun_out/r02a/bench_gmres.json 2> gpurun_out/r02a/bench_gmres.err; echo "bench gmres rc=$?"
timeout 900 python bench.py > gpurun_out/r02a/bench_cg.json 2> gpurun_out/r02a/bench_cg.err; echo "bench cg rc=$?"
timeout 600 python bench.py --force-global --steps 50 --warmup 5 > gpurun_out/r02a/bench_global1.json 2> gpurun_out/r02a/bench_global1.err; echo "bench global rc=$?"
python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r02a/bench_2gpu.out 2>&1; echo "bench --gpus 2 rc=$? (expect 2)"
tail -3 gpurun_out/r02a/shell_tests.log
