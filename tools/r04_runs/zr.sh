# round 4, call 48: persistent workgroups per CU of the box-tile solve (fewer polling waves ahead of the wavefront?)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export RAMD_SLAB_ONLY=gmres
for occ in 0 2 3 4 6; do
  export RAMD_TRSV_WGS_PER_CU=$occ
  echo "wgs/cu=$occ slab64: $(timeout 300 python tools/slab_probe.py 64 2>&1 | grep slab | sed 's/.*iterations//')"
done
for occ in 0 3 4 6; do
  export RAMD_TRSV_WGS_PER_CU=$occ
  echo "wgs/cu=$occ cube: $(timeout 600 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'])")"
done
