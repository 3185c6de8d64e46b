# round 4, call 29: is the first bench process after a heavy job the slow one?  heavy job -> three bench runs; heavy job -> a process that
# allocates and frees most of the device memory -> two bench runs
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zc
mkdir -p $O
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 60 --warmup 10"
heavy() { cd $R; timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_box_tiles_forced.py -m gpu -q > /dev/null 2>&1; cd /tmp; }
bench() { timeout 600 python $R/bench.py $B 2> /dev/null | grep '^{' > $O/line_$1.json; }
export TMPDIR=/tmp
heavy; bench a1; bench a2; bench a3
heavy
python3 - <<'PY'
import ctypes, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','.'))
import rocalution_amd as ra
ra.init_rocalution()
vs=[]
for k in range(15):
    v=ra.LocalVector(); v.Allocate("s", (16<<30)//8); vs.append(v)
ra.sync(); del vs
print("swept 240 GiB")
PY
bench b1; bench b2
heavy; sleep 45; bench c1; bench c2
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04zc/line_*.json')):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms'], 'placement_s', d['placement_s'])
    except Exception as e: print(f, e)
PY
