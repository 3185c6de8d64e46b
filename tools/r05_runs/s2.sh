R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05s2
mkdir -p $O
cd $R
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_rcm.npy timeout 600 python tools/sf_check.py rcm 549 3 > $O/ref_rcm.log 2>&1
run() { tag=$1; shift; env "$@" SF_REF=/tmp/ref_rcm.npy TAG=$tag timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | grep -E "bit-exact|tag=" | sed 's/ilu0.*| LUSolve/LUSolve/' ; }
( run mode3 X=1
  run mode4_far1 RAMD_TRSV_SF_GATHER=4
  run mode4_far0 RAMD_TRSV_SF_GATHER=4 RAMD_TRSV_SF_FAR=0
  run mode4_far2 RAMD_TRSV_SF_GATHER=4 RAMD_TRSV_SF_FAR=2
  run mode4_far1_w4 RAMD_TRSV_SF_GATHER=4 RAMD_TRSV_SF_WAVES=4
  RAMD_TRSV_SF_GATHER=4 RAMD_TRSV_SF_DBG=/tmp/sfdbg TAG=dbg4 timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'
  python tools/sf_timeline.py /tmp/sfdbg_lower.bin
  python tools/sf_timeline.py /tmp/sfdbg_upper.bin ) > $O/knobs.log 2>&1
cat $O/knobs.log
