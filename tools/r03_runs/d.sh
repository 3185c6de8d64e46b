#!/bin/bash
# r03d: placement study -- which offsets between streams change the multi-stream speed (tools/placement2.hip), two processes
O=gpurun_out/r03d; mkdir -p $O
timeout 600 tools/_bin/placement2 > $O/p1.log 2>&1; echo "p1 rc=$?"
timeout 600 tools/_bin/placement2 > $O/p2.log 2>&1; echo "p2 rc=$?"
cat $O/p1.log; echo ======; cat $O/p2.log
