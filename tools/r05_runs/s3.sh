R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05s3
mkdir -p $O
cd $R
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_rcm.npy timeout 600 python tools/sf_check.py rcm 549 3 > $O/ref_rcm.log 2>&1
run() { tag=$1; shift; env "$@" SF_REF=/tmp/ref_rcm.npy TAG=$tag timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | grep -E "bit-exact|tag=" | sed 's/ilu0.*| LUSolve/LUSolve/' ; }
( run default X=1
  run w2 RAMD_TRSV_SF_WAVES=2
  run w4 RAMD_TRSV_SF_WAVES=4
  run w6 RAMD_TRSV_SF_WAVES=6
  run kw2 RAMD_TRSV_SF_KW=2
  run kw4 RAMD_TRSV_SF_KW=4
  run default_again X=1 ) > $O/knobs.log 2>&1
cat $O/knobs.log
