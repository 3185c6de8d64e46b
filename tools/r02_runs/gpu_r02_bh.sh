#!/bin/bash
mkdir -p gpurun_out/r02bh
cd /root/repo
export TMPDIR=/tmp
bash tools/profile_r02.sh > gpurun_out/r02bh/profile.log 2>&1; echo "profile rc=$?"
