# round 4, call 50: the blocked Gram-Schmidt pass on operands chosen by placement class
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout 600 python tools/mgs_class_probe.py 48 2>&1 | grep -v "^$" | tail -30
