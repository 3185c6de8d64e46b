# round 4, call 55: the step records requested in the middle of a step (RAMD_CT_LATE_SMEM=1, the build) against at its start
# (the .late0 library swapped in), alternating: slab, cube, FE surrogate; first the bit-exact suites with the build
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zx
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_box_tiles_forced.py tests/test_gpu_full_size.py tests/test_gpu_shell.py -m gpu -x -q > $O/t.log 2>&1; tail -2 $O/t.log
export RAMD_SLAB_ONLY=gmres
cp rocalution_amd/librocalution_amd.so /tmp/lib_late1.so
run() {
  echo "$1 slab64: $(timeout 300 python tools/slab_probe.py 64 2>&1 | grep slab | sed 's/.*iterations//')"
  echo "$1 cube: $(timeout 600 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 5 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'], 'trsv', d['roofline']['avg_ms'], d['roofline']['min_ms'], d['roofline']['max_ms'])")"
  echo "$1 shell: $(timeout 600 python bench.py --matrix shell --solver gmres --precond ilu0 --steps 100 --warmup 10 --no-cpu-baseline --no-reference-gpu --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('it/s', d['value'], 'trsv', d['roofline']['avg_ms'])")"
}
for rep in 1 2; do
  cp /tmp/lib_late1.so rocalution_amd/librocalution_amd.so; run late1
  cp rocalution_amd/librocalution_amd.so.late0 rocalution_amd/librocalution_amd.so; run late0
done
cp /tmp/lib_late1.so rocalution_amd/librocalution_amd.so
