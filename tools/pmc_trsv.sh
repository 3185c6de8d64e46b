# FETCH_SIZE / WRITE_SIZE of the triangular-solve kernels at 512^3 (one counter per pass); usage: bash tools/pmc_trsv.sh OUTDIR [env...]
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; shift
case $O in /*) ;; *) O=$R/$O;; esac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" timeout 600 rocprofv3 --pmc $c --kernel-trace -d $O/$c -o t -- python $R/tools/trsv_time.py poisson 512 > $O/$c.log 2>&1
  echo "pmc $c rc=$?"
done
python - "$O" <<'PY'
import sqlite3, sys, os
O = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = os.path.join(O, c, "t_results.db")
    if not os.path.exists(db):
        print("missing", db); continue
    rows = sqlite3.connect(db).cursor().execute(
        "select kernel_name,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_trsv_rec%' "
        "or kernel_name like '%k_fill_sentinel%' group by kernel_name,counter_name").fetchall()
    for r in rows:
        print("%s | %s | %d dispatches | avg %.1f KiB" % (r[0].replace("void ramd::", "")[:90], r[1], r[2], r[3]))
        tot.setdefault(r[0], {})[c] = r[3]
n = 512 ** 3
for k, v in tot.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        print("%s: HBM bytes = 2 x FETCH + WRITE = %.3f GB = %.1f B/row = %.2f x algorithmic (62 B/row)" % (k.replace("void ramd::", "")[:60], b / 1e9, b / n, b / n / 62.0))
PY
