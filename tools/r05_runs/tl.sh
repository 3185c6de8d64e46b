# per-unit timestamps of k_trsv_sf on the final kernel (profiles/r05_sf_timeline.txt): hand-off and arithmetic along the critical path
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05tl
mkdir -p $O
cd $R
( for k in rcm delaunay random; do
  echo "== generators.shell_variant(549, '$k'), one LUSolve with RAMD_TRSV_SF_DBG (tools/sf_check.py, tools/sf_timeline.py)"
  RAMD_TRSV_SF_DBG=/tmp/sfdbg_$k TAG=$k timeout 600 python tools/sf_check.py $k 549 2 2>&1 | grep tag= | sed 's/ilu0.*| forms/forms/'
  echo "-- lower"; python tools/sf_timeline.py /tmp/sfdbg_${k}_lower.bin
  echo "-- upper"; python tools/sf_timeline.py /tmp/sfdbg_${k}_upper.bin
done
echo "== without the instrumentation (10 LUSolves each)"
for k in rcm delaunay random; do TAG=$k timeout 600 python tools/sf_check.py $k 549 10 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/'; done
echo "== no dependency waits (RAMD_TRSV_SF_GATHER=2, diagnostic: the stream time)"
RAMD_TRSV_SF_GATHER=2 TAG=rcm_nowait timeout 600 python tools/sf_check.py rcm 549 10 2>&1 | grep tag= | sed 's/ilu0.*| LUSolve/LUSolve/' ) > $O/timeline.txt 2>&1
cat $O/timeline.txt
