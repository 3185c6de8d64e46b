#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_r03.sh > $O/profile.log 2>&1; tail -25 $O/profile.log
python tools/prof_summary.py r03 > $O/summary.log 2>&1; tail -3 $O/summary.log
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_kernel_stats_* profiles/r03_pmc_* gpurun_out/profiles_r03/ 2>/dev/null; ls gpurun_out/profiles_r03 | head -30
