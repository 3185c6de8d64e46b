"""Per-pencil timeline of the 27-point pencil solve (RAMD_TRSV_BOX_DBG): python tools/trsv27_timeline.py N out.txt"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import rocalution_amd as ra
ra.init_rocalution()
N = int(sys.argv[1]); out = sys.argv[2]
A = ra.LocalMatrix(); A.GenLaplace27(N, N, N)
A.ILU0Factorize(); A.LUAnalyse()
n = N ** 3
b = ra.LocalVector(); b.Allocate("", n); b.Ones()
y = ra.LocalVector(); y.Allocate("", n)
for i in range(3):
    A.LUSolve(b, y)
os.environ["RAMD_TRSV_BOX_DBG"] = out
A.LAnalyse(True); A.LSolve(b, y)
a = np.loadtxt(out)
print("pencils", len(a), "end of last (us)", a[:, 5].max())
dur = a[:, 5] - a[:, 4]
print("first-block -> end per pencil: median %.1f min %.1f max %.1f us" % (np.median(dur), dur.min(), dur.max()))
wait = a[:, 4] - a[:, 3]
print("ticket -> first block: median %.1f max %.1f" % (np.median(wait), wait.max()))
print("empty polls per pencil: median %d max %d" % (np.median(a[:, 6]), a[:, 6].max()))
for q in (0, 1, 2, 3, 10, 50, 100, 300, 600, 900, len(a) - 1):
    print("  ", a[q])
# hop lag in J (same K) and K (J+1 -> J at K+1)
d = {(int(r[1]), int(r[2])): r for r in a}
lj = [d[(J + 1, K)][4] - d[(J, K)][4] for (J, K) in d if (J + 1, K) in d]
lk = [d[(J, K + 1)][4] - d[(J + 1, K)][4] for (J, K) in d if (J, K + 1) in d and (J + 1, K) in d]
print("first-block lag (J -> J+1, same K): median %.1f; ((J+1,K) -> (J,K+1)): median %.1f" % (np.median(lj), np.median(lk)))
