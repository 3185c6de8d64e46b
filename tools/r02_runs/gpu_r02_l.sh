#!/bin/bash
# counters of the box-tile triangular solve (512^3): where does a step's time go
mkdir -p gpurun_out/r02l
R=/root/repo
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/r02l/pmc$i -o p --output-format csv -- python $R/bench.py --solver gmres --precond ilu0 --steps 8 --warmup 2 --no-cpu-baseline --no-reference-gpu --no-extras > $R/gpurun_out/r02l/pmc$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python - <<'PY'
import csv,glob,collections,os
out=open('gpurun_out/r02l/summary.txt','w')
for d in sorted(glob.glob('gpurun_out/r02l/pmc*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if 'k_trsv_stream' in k or 'k_mgs_step<double, true>' in k:
                agg[(k[:60],r['Counter_Name'])].append(float(r['Counter_Value']))
        for (k,c),v in sorted(agg.items()):
            out.write("%s | %s | n=%d | avg %.1f\n"%(k,c,len(v),sum(v)/len(v)))
out.close()
print(open('gpurun_out/r02l/summary.txt').read())
PY
