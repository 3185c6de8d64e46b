# round 4, call 41: a rank's slab of the 8-way split (512 x 512 x 64) solved on its own, against an eighth of the full size
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 900 python tools/slab_probe.py 64 2>&1 | grep -v "^$" | tail -8
timeout 900 python tools/slab_probe.py 128 2>&1 | grep "slab" | tail -4
