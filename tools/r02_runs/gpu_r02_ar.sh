#!/bin/bash
cd /root/repo
bash tools/r02_runs/gpu_r02_aq.sh 2>&1 | head -6
bash tools/r02_runs/gpu_r02_an.sh 2>&1 | grep -v "box-tile plan"
