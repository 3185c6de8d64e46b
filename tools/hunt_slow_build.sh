# run the GMRES+ILU(0) bench repeatedly under rocprofv3 --kernel-trace until a slow Build shows up; keep its kernel table
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/hunt
for i in 1 2 3 4 5 6 7 8 9 10; do
  rm -rf /tmp/h$i
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/h$i -o b -- python $R/bench.py --no-cpu-baseline --no-reference-gpu --no-extras --grid 512 --solver gmres --precond ilu0 --steps 6 --warmup 2 --itsolve 3 > /tmp/h$i.out 2>/tmp/h$i.err
  b=$(grep '^{' /tmp/h$i.out | python -c "import json,sys; print(json.load(sys.stdin)['build_s'])")
  echo "run $i build_s $b"
  slow=$(python -c "print(1 if float('$b' or 0) > 3 else 0)")
  if [ "$slow" = "1" ]; then
    python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/h$i/**/b_results.db', recursive=True)[0]
c = sqlite3.connect(db).cursor()
print("top kernels by total:")
for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 12"):
    print("  %s | calls %d | total %.1f ms | avg %.3f ms" % (r[0][:90], r[1], r[2] / 1e6, r[3] / 1e6))
PY
    cp /tmp/h$i/*/b_results.db $R/gpurun_out/hunt/slow_results.db 2>/dev/null || cp $(find /tmp/h$i -name b_results.db | head -1) $R/gpurun_out/hunt/slow_results.db
    break
  fi
done
