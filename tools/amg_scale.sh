cd ${GRAFT_REPO_ROOT:-.}
g++ -std=c++14 -O2 -Iinclude samples/multigrid_driver.cpp -o /tmp/mgd -Lrocalution_amd -lrocalution_amd -Wl,-rpath,$PWD/rocalution_amd
for N in ${1:-32 64 128}; do
  for v in ${2:-c d}; do
    echo "== N=$N variant=$v (c: CG+UAAMG, d: CG+SAAMG)"; timeout 900 /tmp/mgd poisson:$N $v 300 2>&1 | grep "TIMING\|RESULT" | cut -c1-200
  done
done
