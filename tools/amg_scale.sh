cd ${GRAFT_REPO_ROOT:-.}
g++ -std=c++14 -O2 -Iinclude samples/multigrid_driver.cpp -o /tmp/mgd -Lrocalution_amd -lrocalution_amd -Wl,-rpath,$PWD/rocalution_amd
g++ -std=c++14 -O2 -Iinclude samples/krylov_driver.cpp -o /tmp/kd -Lrocalution_amd -lrocalution_amd -Wl,-rpath,$PWD/rocalution_amd
for N in 64 128 256; do
  echo "== N=$N CG+UAAMG"; timeout 600 /tmp/mgd poisson:$N c 300 2>&1 | grep "TIMING\|RESULT" | cut -c1-200
  echo "== N=$N UAAMG solver"; timeout 600 /tmp/mgd poisson:$N a 300 2>&1 | grep "TIMING\|RESULT" | cut -c1-200
done
