R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05s4
mkdir -p $O
cd $R
( RAMD_TRSV_SF=2 RAMD_TRSV_CT=0 RAMD_TRSV_LAT=0 timeout 600 python -m pytest tests/test_gpu_syncfree.py -m gpu -x -q ) > $O/small.log 2>&1; echo "tests rc=$?"; tail -2 $O/small.log
for k in rcm random delaunay; do
RAMD_TRSV_SF=0 SF_SAVE=/tmp/ref_$k.npy timeout 600 python tools/sf_check.py $k 549 3 > $O/ref_$k.log 2>&1
SF_REF=/tmp/ref_$k.npy TAG=$k timeout 600 python tools/sf_check.py $k 549 10 2>&1 | grep -E "bit-exact|tag=" | sed 's/ilu0.*| LUSolve/LUSolve/'
done
RAMD_TRSV_SF_DBG=/tmp/sfdbg TAG=dbg timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'
python tools/sf_timeline.py /tmp/sfdbg_lower.bin
python tools/sf_timeline.py /tmp/sfdbg_upper.bin
