R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
( RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 900 python -m pytest tests/test_gpu_shell.py -m gpu -x -q -k "variants or bit_exact_vs_oracle" ) > $O/small.log 2>&1; echo "small rc=$?"; tail -2 $O/small.log
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_rcm.npy timeout 600 python tools/sf_check.py rcm 549 3 > $O/ref_rcm.log 2>&1
run() { tag=$1; shift; env "$@" SF_REF=/tmp/ref_rcm.npy TAG=$tag timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | grep -E "bit-exact|tag=" | sed 's/ilu0.*| LUSolve/LUSolve/' ; }
( run far1 X=1
  run old RAMD_TRSV_SF_GATHER=3
  run kw6 RAMD_TRSV_SF_KW=6
  run kw3 RAMD_TRSV_SF_KW=3
  run far1_w2 RAMD_TRSV_SF_WAVES=2
  run nodep RAMD_TRSV_SF_GATHER=2
  for g in 0 3; do
  RAMD_TRSV_SF_GATHER=$g RAMD_TRSV_SF_DBG=/tmp/sfdbg$g TAG=dbg$g timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'
  python tools/sf_timeline.py /tmp/sfdbg${g}_lower.bin
  python tools/sf_timeline.py /tmp/sfdbg${g}_upper.bin
  done ) > $O/knobs.log 2>&1
cat $O/knobs.log
