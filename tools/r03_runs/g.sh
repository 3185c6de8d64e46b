#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
export RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_VERBOSE=1
for dd in 0 1; do for gg in 1 0; do
echo "=== dedup=$dd groups=$gg"
RAMD_TRSV_CT_DEDUP=$dd RAMD_TRSV_CT_GROUPS=$gg timeout 300 python tools/r03_runs/g.py 2> $O/err_${dd}_${gg}.log | grep -v "maxdiff 0.0$"
done; done
grep -B3 -A8 "gr3030 cg_ilu1" $O/err_0_1.log | head -40
