# the FIRST bench process after a heavy job runs its fused update kernels at 0.98 instead of 0.86 ms.  Which
# counter differs?  For every counter set: heavy job -> bench under rocprofv3 --pmc (process A: the slow one, if the effect shows)
# -> the same again (process B).  One counter set per pass (gpurun refuses --pmc with the runtime trace domains).
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05fp
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras --steps 30 --warmup 5"
heavy() { cd $R; timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_box_tiles_forced.py -m gpu -q > /dev/null 2>&1; cd /tmp; }
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_avr"; do
  i=$((i+1))
  heavy
  for p in A B; do
    timeout 400 rocprofv3 --pmc $set --kernel-trace -d $O/s${i}$p -o p -- python $R/bench.py $B > $O/line_s${i}$p.json 2> $O/s${i}$p.err
    echo "set $i process $p rc=$?"
  done
done
# one unprofiled pair as the control: is the effect there on this box at all?
heavy
for p in A B C; do timeout 400 python $R/bench.py $B 2>/dev/null | grep '^{' > $O/line_ctl$p.json; done
cd $R && python3 - <<'PY'
import glob, json, os, sqlite3
O = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '.'), 'gpurun_out', 'r05fp')
out = []
for f in sorted(glob.glob(O + '/line_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        out.append('%s: %.1f it/s, spmv %.3f ms, updates %.3f ms' % (os.path.basename(f), d['value'], d['roofline']['avg_ms'], d['kernels']['vector_updates']['avg_ms']))
    except Exception as e:
        out.append('%s: %r' % (f, e))
for db in sorted(glob.glob(O + '/s*/**/*_results.db', recursive=True)):
    try:
        rows = sqlite3.connect(db).cursor().execute(
            "select kernel_name,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_cg_update%' or kernel_name like '%k_cg_direction%' or kernel_name like '%k_csr_pat2%' group by kernel_name,counter_name").fetchall()
        for r in rows:
            out.append('%s | %s | %s | n=%d | avg %.1f' % (db.split('/r05fp/')[1].split('/')[0], r[0][:40], r[1], r[2], r[3]))
    except Exception as e:
        out.append('%s: %r' % (db, e))
open(O + '/summary.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
import shutil
for d in glob.glob(O + '/s*'):
    if os.path.isdir(d):
        shutil.rmtree(d, ignore_errors=True)
PY
