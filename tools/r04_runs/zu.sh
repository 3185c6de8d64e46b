# round 4, call 52: the FE surrogate (grouped box-tile solves, a narrow tile wavefront) with fewer persistent workgroups per CU
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for occ in 0 1 2 4; do
  export RAMD_TRSV_WGS_PER_CU=$occ
  echo "wgs/cu=$occ shell: $(timeout 600 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'], 'trsv avg ms', d['roofline']['avg_ms'])")"
done
