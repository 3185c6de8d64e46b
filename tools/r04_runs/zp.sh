# round 4, call 46: the step record in vector registers (RAMD_CT_RECV=1, the build) against the scalar-cache form (the
# .recv0 library swapped in), alternating: slab, cube, the FE surrogate
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export RAMD_SLAB_ONLY=gmres
cp rocalution_amd/librocalution_amd.so /tmp/lib_recv1.so
run() {
  echo "$1 slab64: $(timeout 300 python tools/slab_probe.py 64 2>&1 | grep slab | sed 's/.*iterations//')"
  echo "$1 cube: $(timeout 600 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'])")"
  echo "$1 shell: $(timeout 600 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'])")"
}
for rep in 1 2; do
  cp /tmp/lib_recv1.so rocalution_amd/librocalution_amd.so; run recv1
  cp rocalution_amd/librocalution_amd.so.recv0 rocalution_amd/librocalution_amd.so; run recv0
done
cp /tmp/lib_recv1.so rocalution_amd/librocalution_amd.so
