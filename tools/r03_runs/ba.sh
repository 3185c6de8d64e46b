#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03ba; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; grep -v "Gloo\|amdgpu.ids" $O/gpu_suite.log | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
