# re-entry check: the whole GPU suite on the restored tree, then ten fresh-process lines right after it
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
cd /tmp; export TMPDIR=/tmp
for i in 0 1 2 3 4 5 6 7 8 9; do
  timeout 400 python $R/bench.py --no-cpu-baseline --no-reference-gpu --no-extras --steps 60 --warmup 10 2>/dev/null | grep '^{' > $O/repeat_$i.json
done
python3 - <<PY > $O/repeats.txt
import json,glob,os
O="$O"
print("# ten fresh processes of the default bench line right after the whole GPU suite (tools/r05_runs/n.sh)")
for f in sorted(glob.glob(O+"/repeat_*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["value"], "it/s  spmv", d["roofline"]["avg_ms"], "updates", d["kernels"]["vector_updates"]["avg_ms"], "placement_s", d["placement_s"])
    except Exception as e: print(f, e)
PY
cat $O/repeats.txt
