#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
timeout 600 tools/_bin/placement2 s > $O/s1.log 2>&1; echo "s1 rc=$?"; cat $O/s1.log
