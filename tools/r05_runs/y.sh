# profiles of the RCM-numbered config-3 class (k_trsv_sf), bench lines of the four numberings, forced sf on the random numbering
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05y
mkdir -p $O
cd $R
bash tools/profile_r05.sh shell_rcm > $O/profile.log 2>&1; tail -3 $O/profile.log
cd /tmp; export TMPDIR=/tmp
for k in rcm delaunay random lex; do
  timeout 900 python $R/bench.py --matrix shell --shell-variant $k --solver gmres --precond ilu0 --steps 60 --warmup 10 --no-cpu-baseline --no-reference-gpu 2> $O/$k.err | grep '^{' > $O/bench_line_shell_$k.json
  echo "$k $(python3 -c "import json;d=json.load(open('$O/bench_line_shell_$k.json'));print(d['value'],'it/s build',d['build_s'],'trsv avg',d['roofline']['avg_ms'],'min',d['roofline']['min_ms'],'max',d['roofline']['max_ms'])")"
done
cd $R
RAMD_TRSV_SF=2 TAG=random_sf timeout 600 python tools/sf_check.py random 549 10 2>&1 | tail -1 | sed 's/ilu0.*| LUSolve/LUSolve/'
RAMD_TRSV_SF=0 TAG=random_levels timeout 600 python tools/sf_check.py random 549 10 2>&1 | tail -1 | sed 's/ilu0.*| LUSolve/LUSolve/'
RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 TAG=lex_sf timeout 600 python tools/sf_check.py lex 549 10 2>&1 | tail -1 | sed 's/ilu0.*| LUSolve/LUSolve/'
