#!/bin/bash
O=gpurun_out/r03cn; mkdir -p $O
export TMPDIR=/tmp
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases.py 512 > $O/build512.log 2>&1; grep "build phase\|GMRES" $O/build512.log | tail -26
RAMD_BUILD_VERBOSE=1 timeout 600 python tools/build_phases_shell.py > $O/build_shell.log 2>&1; grep "GMRES" $O/build_shell.log | tail -3
timeout 3000 python -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
