#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03bl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-reference-gpu --no-extras"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
timeout 600 rocprofv3 --pmc $P1 --kernel-trace -d $O/g1 -o p -- python $R/bench.py --solver gmres --precond ilu0 --steps 12 --warmup 2 $B > $O/g1.log 2>&1; echo rc=$?
timeout 600 rocprofv3 --pmc $P1 --kernel-trace -d $O/c1 -o p -- python $R/bench.py --steps 12 --warmup 2 $B > $O/c1.log 2>&1; echo rc=$?
python - <<PY
import sqlite3,glob
for d in ("$O/g1","$O/c1"):
    dbs=glob.glob(d+"/*.db")+glob.glob(d+"/*/*.db")
    cur=sqlite3.connect(dbs[0]).cursor()
    rows=cur.execute("select kernel_name,counter_name,count(*),avg(value) from counters_collection group by kernel_name,counter_name").fetchall()
    by={}
    for k,c,n,v in rows: by.setdefault(k,{})[c]=(n,v)
    for k,v in sorted(by.items(), key=lambda kv:-kv[1].get('SQ_WAVE_CYCLES',(0,0))[1]*kv[1].get('SQ_WAVE_CYCLES',(0,0))[0])[:9]:
        wc=v.get('SQ_WAVE_CYCLES',(0,1))[1] or 1
        print("%-64s n=%4d  wave_cycles %.3g  wait_any %.2f  wait_inst %.2f  active %.2f  valu %.2f  lds %.2f  waves %.3g  cycles/wave %.0f"%(k.replace('void ramd::','')[:64], v['SQ_WAVE_CYCLES'][0], wc, v['SQ_WAIT_ANY'][1]/wc, v['SQ_WAIT_INST_ANY'][1]/wc, v['SQ_ACTIVE_INST_ANY'][1]/wc, v['SQ_ACTIVE_INST_VALU'][1]/wc, v['SQ_ACTIVE_INST_LDS'][1]/wc, v['SQ_WAVES'][1], 4*wc/max(v['SQ_WAVES'][1],1)))
PY
find $O -name "*.db" -size +8M -delete
