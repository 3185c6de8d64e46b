#!/bin/bash
mkdir -p gpurun_out/r02i
cd /root/repo
export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  for mat in poisson shell; do
    env "$@" RAMD_TRSV_CT_VERBOSE=1 timeout 900 python bench.py --matrix $mat --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu > gpurun_out/r02i/bench_${mat}_$name.json 2> gpurun_out/r02i/bench_${mat}_$name.err; echo "bench $mat $name rc=$?"
  done
}
run nofill_fw0 RAMD_TRSV_NOFILL=1 RAMD_TRSV_CT_FETCHER=0
run nofill_fw1 RAMD_TRSV_NOFILL=1 RAMD_TRSV_CT_FETCHER=1
run d4_fw1 RAMD_TRSV_CT_DEPTH=4 RAMD_TRSV_CT_FETCHER=1
run d8_fw1 RAMD_TRSV_CT_DEPTH=8 RAMD_TRSV_CT_FETCHER=1
run d4_fw0 RAMD_TRSV_CT_DEPTH=4 RAMD_TRSV_CT_FETCHER=0
run old RAMD_TRSV_CT=0
