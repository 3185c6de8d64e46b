#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03bn; mkdir -p $O
export TMPDIR=/tmp
cd $R
RT=$(/opt/rocm/bin/hipcc --print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:verify_asan_link_order=0
env LD_PRELOAD=$RT python -c "print('python under asan ok')" 2>&1 | tail -3
env LD_PRELOAD=$RT RAMD_LIB=$R/rocalution_amd/librocalution_amd_asan.so python -c "
import rocalution_amd as ra, numpy as np
ra.init_rocalution()
A=ra.LocalMatrix(); A.GenPoisson7(32)
x=ra.LocalVector(); x.Allocate('x',32**3); x.Ones(); y=ra.LocalVector(); y.Allocate('y',32**3)
A.Apply(x,y); print('apply ok', y.Norm())
" 2>&1 | tail -8
env LD_PRELOAD=$RT RAMD_LIB=$R/rocalution_amd/librocalution_amd_asan.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "golden or edge or history or io" > $O/asan.log 2>&1; echo "asan subset rc=$?"; tail -6 $O/asan.log; grep -c "AddressSanitizer" $O/asan.log
