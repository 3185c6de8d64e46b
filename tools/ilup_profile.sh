# kernel-time breakdown of an ILU(p) build: tools/ilup_profile.sh N p   (GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P=$R/gpurun_out/prof_ilup
rm -rf $P && mkdir -p $P
timeout 300 rocprofv3 --kernel-trace --stats -d $P/kt -o ilup -- python $R/tools/ilup_time.py $1 $2 > $P/log.txt 2> $P/err.txt < /dev/null
tail -2 $P/log.txt
f=$(find $P/kt -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then head -14 "$f" | cut -c1-150; else echo "no stats file"; tail -5 $P/err.txt; fi
