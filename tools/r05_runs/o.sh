# the sync-free grouped triangular solve (k_trsv_sf): small-size parity with the form forced, then the full-size RCM shell
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
export RAMD_TRSV_CT_VERBOSE=1
( RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 900 python -m pytest tests/test_gpu_shell.py -m gpu -x -q -s -k "variants or bit_exact_vs_oracle" ) > $O/small.log 2>&1; echo "small rc=$?"; tail -15 $O/small.log
for k in rcm delaunay; do
  RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_$k.npy timeout 600 python tools/sf_check.py $k 549 3 > $O/ref_$k.log 2>&1; tail -2 $O/ref_$k.log
  SF_REF=/tmp/ref_$k.npy TAG=sf timeout 600 python tools/sf_check.py $k 549 10 > $O/sf_$k.log 2>&1; tail -4 $O/sf_$k.log
done
for w in 1 2 4; do RAMD_TRSV_SF_WAVES=$w TAG=waves$w timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | tail -1; done > $O/waves.log 2>&1; cat $O/waves.log
for c in 1 4 32; do RAMD_TRSV_SF_POLLCAP=$c TAG=pollcap$c timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | tail -1; done > $O/pollcap.log 2>&1; cat $O/pollcap.log
