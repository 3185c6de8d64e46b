"""ad-hoc first-look timings (wall clock around N back-to-back launches + device sync)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra

ra.init_rocalution()
print(ra.info_rocalution())
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = 50
A = ra.LocalMatrix(); t0 = time.time(); A.GenPoisson7(N); ra.sync(); print("gen %.3fs" % (time.time() - t0))
n, nnz = A.GetM(), A.GetNnz()
x = ra.LocalVector(); x.Allocate("", n); x.Ones()
y = ra.LocalVector(); y.Allocate("", n)
def bench(f, bytes_, name):
    for _ in range(5): f()
    ra.sync(); t0 = time.time()
    for _ in range(reps): f()
    ra.sync(); dt = (time.time() - t0) / reps
    print("%-14s %8.3f ms  %8.1f GB/s" % (name, dt * 1e3, bytes_ / dt / 1e9)); return dt
bench(lambda: A.Apply(x, y), 4 * (n + nnz) + 8 * (2 * n + nnz), "csr spmv")
print("check y sum", y.Reduce(), "expected", float(6 * N * N))
bench(lambda: y.AddScale(x, 0.5), 24 * n, "axpy")
bench(lambda: y.Dot(x), 16 * n, "dot(blocking)")
bench(lambda: y.ScaleAdd2(0.5, x, 0.25, x, 0.1), 32 * n, "scaleadd2")
E = ra.LocalMatrix(); E.CloneFrom(A); E.ConvertToELL()
bench(lambda: E.Apply(x, y), 4 * 7 * n + 8 * (2 * n + 7 * n), "ell spmv")
H = ra.LocalMatrix(); H.CloneFrom(A); H.ConvertToHYB()
bench(lambda: H.Apply(x, y), 4 * 7 * n + 8 * (2 * n + 7 * n), "hyb spmv")
t0 = time.time(); LU = ra.LocalMatrix(); LU.CloneFrom(A); LU.ILU0Factorize(); ra.sync(); print("ilu0 %.3fs" % (time.time() - t0))
t0 = time.time(); LU.LUAnalyse(); ra.sync(); print("lu analyse %.3fs" % (time.time() - t0))
bench(lambda: LU.LUSolve(x, y), 2 * (12 * nnz) , "lusolve")
