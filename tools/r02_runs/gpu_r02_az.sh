#!/bin/bash
cd /root/repo
timeout 1500 python tools/stress_trsv.py 256 1500 2>&1 | tail -4
