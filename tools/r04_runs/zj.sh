# round 4, call 38: Build() time of the AMG on the GlobalMatrix at 256^3, one rank against four ranks sharing the device
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zj
mkdir -p $O
cd $R
for pc in global-saamg global-uaamg; do
  for g in 1 4; do
    S=$SECONDS
    timeout 1200 python bench.py --gpus $g --transport callback --grid 256 --precond $pc --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras 2>$O/${pc}_$g.err | grep '^{' > $O/${pc}_$g.json
    echo "$pc gpus=$g wall=$((SECONDS-S))s $(python -c "
import json
try:
    d=json.loads(open('$O/${pc}_$g.json').read()); print('it/s', d['value'], 'build_s', d.get('build_s'), 'final', d.get('final_residual'))
except Exception as e:
    print('no line', e)
")"
  done
done
