R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && g++ -std=c++14 -O2 -Iinclude samples/multigrid_driver.cpp -o /tmp/mgd -Lrocalution_amd -lrocalution_amd -Wl,-rpath,$R/rocalution_amd
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/amgprof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/amgprof -o p -- /tmp/mgd poisson:${1:-256} ${2:-d} 300 2>&1 | grep "TIMING"
python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/amgprof/**/p_results.db', recursive=True)[0]
c = sqlite3.connect(db).cursor()
for r in c.execute("select name,total_calls,total_duration,average from top_kernels order by total_duration desc limit 14"):
    print("%-90s calls %6d total %9.1f ms avg %9.3f ms" % (r[0][:90], r[1], r[2] / 1e6, r[3] / 1e6))
PY
