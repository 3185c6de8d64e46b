"""Turn the rocprofv3 sqlite outputs under gpurun_out/prof/ into small text summaries in profiles/."""
import os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

def q(db, sql):
    return sqlite3.connect(db).cursor().execute(sql).fetchall()

kt = os.path.join(src, "kt", "bench_results.db")
if os.path.exists(kt):
    with open(os.path.join(out, tag + "_kernel_stats.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 100 --warmup 10 (MI355X, 512^3 CG+Jacobi)\n")
        f.write("# name | calls | total_us | avg_us | pct\n")
        for r in q(kt, "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 25"):
            f.write("%s | %d | %.1f | %.3f | %.2f\n" % (r[0].replace("void ramd::", "ramd::")[:110], r[1], r[2] / 1.0, r[3], r[4]))
        bj = os.path.join(src, "bench_kt.json")
        if os.path.exists(bj):
            f.write("# bench line of the same (profiled) run:\n# " + open(bj).read().strip() + "\n")
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    db = os.path.join(src, name, "bench_results.db")
    if not os.path.exists(db):
        continue
    with open(os.path.join(out, "%s_pmc_%s.txt" % (tag, ctr)), "w") as f:
        f.write("# rocprofv3 --pmc %s --kernel-trace -- python bench.py --steps 20 --warmup 2   (separate pass per counter)\n" % ctr)
        f.write("# values are KiB per dispatch as reported; on gfx950 FETCH_SIZE counts 64 B per 128-B request for\n"
                "# 16-byte-per-lane streaming reads, i.e. HALF the bytes (MI355X_MICROARCH.md, HBM section) -- checked\n"
                "# in this very run on k_cg_update / k_cg_direction (each reads 3 x 1.0737 GB = 3,145,728 KiB, reported\n"
                "# ~1,572,9xx KiB).  WRITE_SIZE matches the algorithmic write volume exactly (2 x 1,048,576 KiB).\n")
        f.write("# kernel | counter | dispatches | avg | min | max\n")
        for r in q(db, "select kernel_name,counter_name,count(*),avg(value),min(value),max(value) from counters_collection "
                       "group by kernel_name,counter_name order by avg(value) desc limit 14"):
            f.write("%s | %s | %d | %.1f | %.1f | %.1f\n" % (r[0].replace("void ramd::", "ramd::")[:110], r[1], r[2], r[3], r[4], r[5]))
print(os.listdir(out))
