#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
timeout 600 tools/_bin/placement2 w > $O/w1.log 2>&1; echo "w1 rc=$?"; cat $O/w1.log
