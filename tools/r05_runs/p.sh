# k_trsv_sf: entries per lane, the single-word wait against gathering every turn, the no-wait floor
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
( RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 900 python -m pytest tests/test_gpu_shell.py -m gpu -x -q -k "variants or bit_exact_vs_oracle" ) > $O/small.log 2>&1; echo "small rc=$?"; tail -2 $O/small.log
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_rcm.npy timeout 600 python tools/sf_check.py rcm 549 3 > $O/ref_rcm.log 2>&1
run() { tag=$1; shift; env "$@" SF_REF=/tmp/ref_rcm.npy TAG=$tag timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | grep -E "bit-exact|tag=" | sed 's/ilu0.*| LUSolve/LUSolve/' ; }
( run default X=1
  run kw2 RAMD_TRSV_SF_KW=2
  run kw6 RAMD_TRSV_SF_KW=6
  run gather RAMD_TRSV_SF_GATHER=1
  run gather_cap1 RAMD_TRSV_SF_GATHER=1 RAMD_TRSV_SF_POLLCAP=1
  run gather_w2 RAMD_TRSV_SF_GATHER=1 RAMD_TRSV_SF_WAVES=2
  run gather_w1 RAMD_TRSV_SF_GATHER=1 RAMD_TRSV_SF_WAVES=1
  run nodep RAMD_TRSV_SF_GATHER=2
  run nodep_w2 RAMD_TRSV_SF_GATHER=2 RAMD_TRSV_SF_WAVES=2 ) > $O/knobs.log 2>&1
cat $O/knobs.log
