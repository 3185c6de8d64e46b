# round 4, call 42: kernel times of GMRES(30)+ILU(0) on the 512 x 512 x 64 slab
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zm
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
RAMD_SLAB_ONLY=gmres timeout 900 rocprofv3 --kernel-trace --stats -d $O/slab64 -o slab -- python $R/tools/slab_probe.py 64 > $O/run.log 2>&1
grep slab $O/run.log | tail -3
python $R/tools/db_summary.py $O
head -16 $O/slab64.txt
