# k_trsv_sf<XL>: one XCD, hand-offs through its L2
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05u
mkdir -p $O
cd $R
export RAMD_TRSV_SF_XL=1
( RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 600 python -m pytest tests/test_gpu_shell.py -m gpu -x -q -k "variants or bit_exact_vs_oracle" ) > $O/small.log 2>&1; echo "small rc=$?"; tail -2 $O/small.log
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_rcm.npy timeout 600 python tools/sf_check.py rcm 549 3 > $O/ref_rcm.log 2>&1
run() { tag=$1; shift; env "$@" SF_REF=/tmp/ref_rcm.npy TAG=$tag timeout 300 python tools/sf_check.py rcm 549 10 2>&1 | grep -E "bit-exact|tag=|rror|abort" | sed 's/ilu0.*| LUSolve/LUSolve/' ; }
( run xl_w16 X=1
  run xl_w8 RAMD_TRSV_SF_WAVES=8
  run xl_w4 RAMD_TRSV_SF_WAVES=4
  run xl_w16_gather RAMD_TRSV_SF_GATHER=1
  run xl_w16_gather_cap1 RAMD_TRSV_SF_GATHER=1 RAMD_TRSV_SF_POLLCAP=1
  run xl_w16_far RAMD_TRSV_SF_GATHER=0
  run noxl RAMD_TRSV_SF_XL=0
  RAMD_TRSV_SF_DBG=/tmp/sfdbg TAG=dbg timeout 300 python tools/sf_check.py rcm 549 2 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'
  python tools/sf_timeline.py /tmp/sfdbg_lower.bin
  python tools/sf_timeline.py /tmp/sfdbg_upper.bin ) > $O/knobs.log 2>&1
cat $O/knobs.log
