# round 4, call 14: the default bench line end to end with its wall time; the placement class over the device memory
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SECONDS=0
timeout 1500 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$? wall ${SECONDS}s"
grep -E "reference_gpu" $O/bench_default.err | head
for rep in 1 2; do
  timeout 600 python $R/tools/class_map.py 4 24 2>&1 | grep -v "^rocal" > $O/class_map_4_$rep.log; tail -1 $O/class_map_4_$rep.log
done
timeout 600 python $R/tools/class_map.py 1 200 2>&1 | grep -v "^rocal" > $O/class_map_1.log; tail -1 $O/class_map_1.log
timeout 600 python $R/tools/class_map.py 16 24 2>&1 | grep -v "^rocal" > $O/class_map_16.log; tail -1 $O/class_map_16.log
