"""Summarise every rocprofv3 sqlite output below a directory into a text file next to it (kernel stats and, where a counter
pass was made, the per-kernel averages of every counter), then delete the databases: gpurun merges at most 64 MiB back."""
import glob
import os
import shutil
import sqlite3
import sys

top = sys.argv[1]
for db in sorted(glob.glob(os.path.join(top, "*", "**", "*_results.db"), recursive=True)):
    rel = os.path.relpath(db, top).split(os.sep)[0]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    with open(os.path.join(top, rel + ".txt"), "w") as f:
        if "top_kernels" in tables:
            f.write("# name | calls | total_ns | avg_ns | pct\n")
            for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 40"):
                f.write("%s | %d | %.0f | %.1f | %.2f\n" % (r[0][:140], r[1], r[2], r[3], r[4]))
        if "counters_collection" in tables:
            f.write("# kernel | counter | dispatches | avg | min | max\n")
            for r in cur.execute("select kernel_name,counter_name,count(*),avg(value),min(value),max(value) from counters_collection "
                                 "group by kernel_name,counter_name order by kernel_name,counter_name"):
                if r[3] and r[3] > 0:
                    f.write("%s | %s | %d | %.1f | %.1f | %.1f\n" % (r[0][:140], r[1], r[2], r[3], r[4], r[5]))
    con.close()
    shutil.rmtree(os.path.join(top, rel), ignore_errors=True)
