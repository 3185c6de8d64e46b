# round 4, call 53: instruction counts of the box-tile solve on the slab (how many instructions does a step cost?)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zv
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RAMD_SLAB_ONLY=gmres RAMD_TRSV_CT_VERBOSE=1
for ctr in "SQ_INSTS_SALU SQ_INSTS_VALU" "SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $ctr | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/$tag -o slab -- python $R/tools/slab_probe.py 64 > $O/run_$tag.log 2>&1
done
grep "box-tile plan" $O/run_SQ_INSTS_SALU_SQ_INSTS_VALU.log | head -8
python $R/tools/db_summary.py $O
grep -h "k_trsv_rec" $O/*.txt | cut -c1-60,150-260
