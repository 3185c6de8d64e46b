"""a few SpMV launches at N^3 for PMC passes:  python tools/spmv_only.py [N] [reps] [format] [poisson|lap27]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fmt = sys.argv[3] if len(sys.argv) > 3 else "csr"
ra.init_rocalution()
which = sys.argv[4] if len(sys.argv) > 4 else "poisson"
A = ra.LocalMatrix()
if which == "lap27":
    A.GenLaplace27(N)
else:
    A.GenPoisson7(N)
n = N ** 3
x = ra.LocalVector(); x.Allocate("x", n); x.Ones()
y = ra.LocalVector(); y.Allocate("y", n)
if fmt != "csr":
    A.ConvertTo({"ell": ra.ELL, "hyb": ra.HYB}[fmt])
for _ in range(reps):
    A.Apply(x, y)
ra.sync()
print("done", y.Norm())
