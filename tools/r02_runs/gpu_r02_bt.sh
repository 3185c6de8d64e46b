#!/bin/bash
# staggered big allocations (RAMD_ALLOC_STAGGER bytes x (k mod 16)): does the spread of the CG update over vector groups close?
mkdir -p gpurun_out/r02bt
cd /root/repo
export TMPDIR=/tmp
for st in 0 256 4096 69632 1114112 0 4096; do
RAMD_ALLOC_STAGGER=$st timeout 300 python tools/placement_probe.py 14 > gpurun_out/r02bt/p_$st.log 2>&1; echo "stagger=$st rc=$?"; grep "cg_update" gpurun_out/r02bt/p_$st.log | python -c "
import sys,re
for l in sys.stdin:
    v=[int(x) for x in re.findall(r'\) (\d+)', l)]
    print('   groups:', ' '.join(str(x) for x in v[:-1]), ' min %d max %d mean %d'%(min(v[:-1]),max(v[:-1]),sum(v[:-1])/len(v[:-1])))"
done
