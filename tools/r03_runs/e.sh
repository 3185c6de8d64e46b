#!/bin/bash
# r03e: 14 vectors inside one allocation at controlled strides, three processes: are all groups fast and is it reproducible?
O=gpurun_out/r03e; mkdir -p $O
for i in 1 2 3; do timeout 600 tools/_bin/placement2 arena > $O/p$i.log 2>&1; echo "p$i rc=$?"; grep arena $O/p$i.log; done
