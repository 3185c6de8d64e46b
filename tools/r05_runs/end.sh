# end of the round, on the final tree: the whole GPU suite, the smoke test, the default bench line as the driver runs it, the
# mixed-precision line again (its columns-read leg measured the pattern kernel until the cast copy kept the switch)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05end
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
cd /tmp; export TMPDIR=/tmp
timeout 900 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2> $O/cg_end.err | grep '^{' > $O/bench_line_cg_end.json; python3 -c "import json;d=json.load(open('$O/bench_line_cg_end.json'));print('default line', d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], (d.get('roofline_columns_read') or {}).get('frac'))"
timeout 900 python $R/bench.py --solver mixed --steps 30 --warmup 3 2> $O/mixed.err | grep '^{' > $O/bench_line_mixed.json; python3 -c "import json;d=json.load(open('$O/bench_line_mixed.json'));print('mixed', d['value'], d['roofline']['avg_ms'], d['roofline']['frac'], (d.get('roofline_columns_read') or {}).get('avg_ms'), (d.get('roofline_columns_read') or {}).get('frac'))"
