R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05t
mkdir -p $O
cd $R
( RAMD_TRSV_SF_DBG=/tmp/sfdbg TAG=dbg timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'
  python tools/sf_timeline.py /tmp/sfdbg_lower.bin
  python tools/sf_timeline.py /tmp/sfdbg_upper.bin ) > $O/timeline.log 2>&1
cat $O/timeline.log
