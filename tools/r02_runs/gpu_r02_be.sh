#!/bin/bash
mkdir -p gpurun_out/r02be
cd /root/repo
export TMPDIR=/tmp
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_DEDUP=1 timeout 3000 python -m pytest tests -q -m gpu --deselect tests/test_gpu_box_tiles_forced.py > gpurun_out/r02be/forced_all.log 2>&1; echo "whole suite, box tiles forced rc=$?"; tail -8 gpurun_out/r02be/forced_all.log | cut -c1-250
