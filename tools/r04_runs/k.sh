# round 4, call 11: the whole GPU suite with the final code, then the default bench line (twice), then the profile passes
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04k
mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 1200 python $R/bench.py > $O/bench_default_1.json 2> $O/bench_default_1.err; echo "bench default rc=$?"
timeout 1200 python $R/bench.py --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; echo "bench driver-like rc=$?"
RAMD_CSR_PAT=0 timeout 600 python $R/bench.py --no-cpu-baseline --no-reference-gpu --no-extras --solver mixed --steps 30 --warmup 3 > $O/line_mixed_columns_read.json 2> /dev/null; echo "mixed columns-read rc=$?"
cd $R && bash tools/profile_r04.sh > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
