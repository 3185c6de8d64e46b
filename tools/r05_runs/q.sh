# k_trsv_sf: the single-word wait on a position a few levels back (depth 0..3) against the wait on the last dependency
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
( RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 900 python -m pytest tests/test_gpu_shell.py -m gpu -x -q -k "variants or bit_exact_vs_oracle" ) > $O/small.log 2>&1; echo "small rc=$?"; tail -2 $O/small.log
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_rcm.npy timeout 600 python tools/sf_check.py rcm 549 3 > $O/ref_rcm.log 2>&1
run() { tag=$1; shift; env "$@" SF_REF=/tmp/ref_rcm.npy TAG=$tag timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | grep -E "bit-exact|tag=" | sed 's/ilu0.*| LUSolve/LUSolve/' ; }
( run far1 X=1
  run far0 RAMD_TRSV_SF_FAR=0
  run far2 RAMD_TRSV_SF_FAR=2
  run far3 RAMD_TRSV_SF_FAR=3
  run far1_w2 RAMD_TRSV_SF_WAVES=2
  run far1_w4 RAMD_TRSV_SF_WAVES=4
  run far2_w4 RAMD_TRSV_SF_WAVES=4 RAMD_TRSV_SF_FAR=2
  run old RAMD_TRSV_SF_GATHER=3 ) > $O/knobs.log 2>&1
cat $O/knobs.log
