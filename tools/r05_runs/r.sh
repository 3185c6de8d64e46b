R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
for g in 0 3; do
RAMD_TRSV_SF_GATHER=$g RAMD_TRSV_SF_DBG=/tmp/sfdbg$g TAG=dbg$g timeout 600 python tools/sf_check.py rcm 549 2 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'
python tools/sf_timeline.py /tmp/sfdbg${g}_lower.bin
python tools/sf_timeline.py /tmp/sfdbg${g}_upper.bin
done > $O/timeline.log 2>&1
cat $O/timeline.log
