# PMC passes over one kernel of one command: where do its waves spend their cycles?  (generalised from pmc_spmv_shell.sh, round 6)
# usage (GPU box, repo root): bash tools/pmc_kernel.sh OUTDIR KERNEL_SUBSTRING -- command ...   -> OUTDIR/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$1; K=$2; shift 3
case $O in /*) ;; *) O=$R/$O;; esac; rm -rf $O; mkdir -p $O
CMD="$@"
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; ( cd $R && timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/$tag -o p -- $CMD ) > $O/$tag.log 2>&1; echo "$tag rc=$?"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT
run p3 TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY
run p4 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES
run p5 TCC_HIT TCC_MISS TCC_REQ
run p6 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
run p7 FETCH_SIZE
run p8 WRITE_SIZE
python - <<PY
import sqlite3,glob,os
out=open("$O/summary.txt","w")
out.write("# rocprofv3 --pmc ... --kernel-trace -- $CMD ; kernels matching '$K'; per-dispatch averages as reported (FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE counts 64 B per 128-B request on gfx950)\n")
for d in sorted(glob.glob("$O/p?")):
    dbs=glob.glob(d+"/*.db")+glob.glob(d+"/*/*.db")
    if not dbs: out.write("%s: no db\n"%d); continue
    cur=sqlite3.connect(dbs[0]).cursor()
    try:
        tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
        t=[x for x in tabs if x.startswith("counters_collection")][0]
        rows=cur.execute("select kernel_name,counter_name,count(*),avg(value) from %s where kernel_name like '%%$K%%' group by kernel_name,counter_name"%t).fetchall()
    except Exception as e:
        out.write("%s: %s\n"%(d,e)); continue
    for r in rows:
        out.write("%s | %s | %s | n=%d | avg %.6g\n"%(os.path.basename(d), r[0].replace("void ramd::","")[:60], r[1], r[2], r[3]))
out.close()
print(open("$O/summary.txt").read())
PY
find $O -name "*.db" -size +8M -delete; find $O -name "*.csv" -size +1M -delete
