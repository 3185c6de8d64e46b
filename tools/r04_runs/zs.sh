# round 4, call 49: in-kernel phase timers of the box-tile solves at 512^3 and on the slab (RAMD_TRSV_PROF=1)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zs
mkdir -p $O
cd $R
RAMD_TRSV_PROF=1 timeout 600 python bench.py --solver gmres --precond ilu0 --steps 20 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras > /tmp/c.log 2>&1; echo "cube rc=$?"
grep "trsv prof" /tmp/c.log | tail -8 > $O/cube.txt; cat $O/cube.txt
RAMD_SLAB_ONLY=gmres RAMD_TRSV_PROF=1 timeout 120 python tools/slab_probe.py 64 > /tmp/p.log 2>&1
grep "trsv prof" /tmp/p.log | tail -4 > $O/slab.txt; cat $O/slab.txt
