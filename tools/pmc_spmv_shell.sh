# PMC passes over the CSR SpMV of the config-3 surrogate (tools/spmv_shell.py): where do the 35-entry rows spend their cycles?
# usage (GPU box, repo root): bash tools/pmc_spmv_shell.sh OUTDIR   -> OUTDIR/summary.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=${1:-gpurun_out/pmc_spmv_shell}; case $O in /*) ;; *) O=$R/$O;; esac; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $O/$tag -o p -- python $R/tools/spmv_shell.py 549 > $O/$tag.log 2>&1; echo "$tag rc=$?"; }
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run p2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT
run p3 TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY
run p4 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES
run p5 TCC_HIT TCC_MISS TCC_REQ
run p6 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE
python - <<PY
import sqlite3,glob,os
out=open("$O/summary.txt","w")
out.write("# rocprofv3 --pmc ... --kernel-trace -- python tools/spmv_shell.py 549 (config-3 surrogate, k_csr_tr general form), per-dispatch averages over all XCDs/SEs as reported\n")
for d in sorted(glob.glob("$O/p?")):
    dbs=glob.glob(d+"/*.db")+glob.glob(d+"/*/*.db")
    if not dbs: out.write("%s: no db\n"%d); continue
    cur=sqlite3.connect(dbs[0]).cursor()
    try:
        rows=cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection where kernel_name like '%k_csr_tr%' group by kernel_name,counter_name").fetchall()
    except Exception as e:
        out.write("%s: %s\n"%(d,e)); continue
    for r in rows:
        out.write("%s | %s | %s | n=%d | avg %.5g\n"%(os.path.basename(d), r[0].replace("void ramd::","")[:44], r[1], r[2], r[3]))
out.close()
print(open("$O/summary.txt").read())
PY
find $O -name "*.db" -size +8M -delete; find $O -name "*.csv" -size +1M -delete
