# round 4, call 13: plan-identity regression test, the default bench line end to end (with the vendor columns of all legs)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04m
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_distributed.py -m gpu -q -x -k "coincide or rehears" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
/usr/bin/time -v timeout 1500 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
grep -E "Elapsed|reference_gpu" $O/bench_default.err | head
