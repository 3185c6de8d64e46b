R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
RAMD_BUILD_VERBOSE=1 RAMD_TRSV_CT_VERBOSE=1 TAG=build timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep -v "^$" | tail -60 | cut -c1-200
