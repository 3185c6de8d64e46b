#!/bin/bash
# kernel trace of the GMRES(30)+ILU(0) 512^3 bench with the blocked MGS (where do the 18.3 ms of an iteration go)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r02bm
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02bm/kt -o bench -- python $R/bench.py --no-cpu-baseline --no-reference-gpu --no-extras --solver gmres --precond ilu0 --steps 60 --warmup 10 > $R/gpurun_out/r02bm/bench.json 2> $R/gpurun_out/r02bm/err.log
echo rc=$?
f=$(find $R/gpurun_out/r02bm/kt -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(r["Name"][:90], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
find $R/gpurun_out/r02bm/kt -name "*.db" -size +30M -delete
