# PMC passes over the CSR SpMV alone (tools/spmv_time.py): where does the row-pattern product spend its cycles?
# usage (GPU box, repo root): bash tools/pmc_spmv.sh OUTDIR   -> OUTDIR/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=${1:-gpurun_out/pmc_spmv}; case $O in /*) ;; *) O=$R/$O;; esac; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # tag, env..., then counters
  tag=$1; shift
  envs=""
  while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done
  env $envs timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/$tag -o p -- python $R/tools/spmv_time.py 512 20 > $O/$tag.log 2>&1
  echo "$tag rc=$?"
}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT"
P3="TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY"
P4="TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES"
P5="TCC_HIT TCC_MISS TCC_REQ"
P6="TCP_TCR_TCP_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES"
for v in pat cols; do
  E="DOT=1"; [ $v = cols ] && E="DOT=1 RAMD_CSR_PAT=0"
  run ${v}_1 $E $P1; run ${v}_2 $E $P2; run ${v}_3 $E $P3; run ${v}_4 $E $P4; run ${v}_5 $E $P5; run ${v}_6 $E $P6
done
python - <<PY
import sqlite3,glob,os
out=open("$O/summary.txt","w")
for d in sorted(glob.glob("$O/*_?")):
    dbs=glob.glob(d+"/*.db")+glob.glob(d+"/*/*.db")
    if not dbs: out.write("%s: no db\n"%d); continue
    cur=sqlite3.connect(dbs[0]).cursor()
    try:
        rows=cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_csr_tr%' group by kernel_name,counter_name").fetchall()
    except Exception as e:
        out.write("%s: %s\n"%(d,e)); continue
    for r in rows:
        out.write("%s | %s | %s | n=%d | avg %.4g\n"%(os.path.basename(d), r[0][:60], r[1], r[2], r[3]))
out.close()
print(open("$O/summary.txt").read())
PY
find $O -name "*.db" -size +8M -delete; find $O -name "*.csv" -size +1M -delete
