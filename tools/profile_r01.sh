# rocprofv3 passes behind profiles/r01_* (run on the GPU box from the repo root; then: python tools/prof_summary.py r01)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof
rm -rf $P && mkdir -p $P
B="--no-cpu-baseline --no-reference-gpu"
timeout 900 rocprofv3 --kernel-trace --stats -d $P/kt -o bench -- python $R/bench.py --steps 100 --warmup 10 $B > $P/bench_kt.json 2> $P/kt.err
grep '^{' $P/bench_kt.json > $P/bench_kt.json.tmp; mv $P/bench_kt.json.tmp $P/bench_kt.json
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $P/fetch -o bench -- python $R/bench.py --steps 20 --warmup 2 $B --no-extras > /dev/null 2> $P/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $P/write -o bench -- python $R/bench.py --steps 20 --warmup 2 $B --no-extras > /dev/null 2> $P/write.err
ls $P/kt $P/fetch $P/write
