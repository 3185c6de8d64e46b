# round 4, call 47: the in-kernel phase timers of the box-tile solve on the slab (RAMD_TRSV_PROF=1)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export RAMD_SLAB_ONLY=gmres
RAMD_TRSV_PROF=1 timeout 120 python tools/slab_probe.py 64 > /tmp/p.log 2>&1; echo "rc=$?"
grep -c "trsv prof" /tmp/p.log; grep "trsv prof" /tmp/p.log | sed -n '20,23p'; grep -v "trsv prof" /tmp/p.log | tail -5
