R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for w in 0 1 2 3 4 6; do RAMD_TRSV_WGS_PER_CU=$w TAG=wgs$w timeout 600 python tools/trsv_time.py shell 549 2>&1 | tail -1 | sed 's/ilu0.*| LUSolve/LUSolve/'; done
