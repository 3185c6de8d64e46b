R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
RAMD_TRSV_CT=0 TAG=lex_sf timeout 600 python tools/sf_check.py lex 549 10 2>&1 | grep tag= | sed 's/ilu0.*| forms/forms/' | cut -c1-250
RAMD_TRSV_CT=0 RAMD_TRSV_SF_WAVES=8 TAG=lex_sf_w8 timeout 600 python tools/sf_check.py lex 549 10 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/' | cut -c1-250
TAG=lex_tiles timeout 600 python tools/sf_check.py lex 549 10 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/' | cut -c1-250
