R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
RAMD_TRSV_CT_VERBOSE=1 TAG=morton timeout 600 python tools/sf_check.py morton 549 10 2>&1 | grep -E "plan|tag=" | cut -c1-260
cd /tmp; export TMPDIR=/tmp
timeout 900 python $R/bench.py --matrix shell --shell-variant morton --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu 2>/dev/null | grep '^{' > $R/gpurun_out/bench_line_shell_morton.json
python3 -c "import json;d=json.load(open('$R/gpurun_out/bench_line_shell_morton.json'));print('morton', d['value'], d['ms_per_step'], d['build_s'], d['roofline']['min_ms'], d['roofline']['max_ms'], d['tri_plan']['lower']['form'][:40], d['tri_plan']['lower']['dependency_levels'])"
