# round 4, call 36: CG + UA-AMG on the GlobalMatrix, residual after 10 iterations on 1 / 2 / 4 ranks, twice each
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zi
mkdir -p $O
cd $R
for g in 1 1 2 4 4; do
  timeout 600 python bench.py --gpus $g --transport callback --grid 64 --precond global-uaamg --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gpus', $g, 'final', d['final_residual'], 'it/s', d['value'])"
done
