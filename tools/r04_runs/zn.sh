# round 4, call 44: cubic tile boxes of the triangular solves on the 512 x 512 x 64 slab and, two of them, on the full cube
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export RAMD_SLAB_ONLY=gmres
for box in 8,8,8 10,10,10 12,12,12 16,16,16 16,16,8 12,12,8 8,8,16 6,6,6 10,10,8; do
  echo "box=$box: $(RAMD_TRSV_CT_BOX=$box timeout 300 python tools/slab_probe.py 64 2>&1 | grep slab)"
done
for box in default 10,10,10 12,12,12 16,16,16; do
  if [ $box = default ]; then unset RAMD_TRSV_CT_BOX; else export RAMD_TRSV_CT_BOX=$box; fi
  echo "cube box=$box: $(timeout 600 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'], {k:round(v['avg_ms'],3) for k,v in d.get('kernels',{}).items() if 'avg_ms' in v})")"
done
