# kernel-time breakdown of an ILU(p) build: tools/ilup_profile.sh N p   (GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof_ilup
rm -rf $P && mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats -d $P/kt -o ilup -- python $R/tools/ilup_time.py $1 $2 > $P/log.txt 2> $P/err.txt < /dev/null
tail -2 $P/log.txt
# rocprofv3 of this image writes a rocpd database: per-kernel totals from it
python - "$P/kt" <<'PY'
import glob, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)
if not db:
    raise SystemExit("no rocpd database under " + sys.argv[1])
cur = sqlite3.connect(db[0]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
for r in cur.execute("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6 from %s d join %s s "
                     "on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit 16" % (kd, ks)):
    print("%-98s %8d %12.3f %12.4f" % (r[0][:98], r[1], r[2], r[3]))
PY
