cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 ${1:-10}); do echo "proc $i"; timeout 300 python tools/hunt_all.py 512 > /tmp/hunt.out 2>&1; rc=$?; grep build /tmp/hunt.out; if [ $rc != 0 ]; then echo "rc=$rc"; tail -5 /tmp/hunt.out; fi; done
