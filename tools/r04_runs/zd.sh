# round 4, call 30: free device memory second by second right after a heavy job ended
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04zd
mkdir -p $O
cd $R; timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_box_tiles_forced.py -m gpu -q > /dev/null 2>&1
cd /tmp; timeout 200 python $R/tools/mem_after.py 50 2>&1 | grep "^t=" > $O/mem.log
awk 'NR<=12 || NR%10==0' $O/mem.log
rocm-smi --showmemuse --showuse 2>/dev/null | head -20
