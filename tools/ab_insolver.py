"""same process, same box: SpMV back-to-back vs inside the CG loop vs interleaved with vector kernels"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import capi, solvers as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ra.init_rocalution()
lib = capi.load()
n = N ** 3
A = ra.LocalMatrix(); A.GenPoisson7(N)
x = ra.LocalVector(); x.Allocate("x", n); x.Ones()
y = ra.LocalVector(); y.Allocate("y", n)
z = ra.LocalVector(); z.Allocate("z", n); z.Ones()
w = ra.LocalVector(); w.Allocate("w", n); w.Ones()


def prof(f, reps):
    capi.check(lib.ramd_prof_spmv_enable(1))
    for _ in range(reps):
        f()
    ra.sync()
    cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
    capi.check(lib.ramd_prof_spmv_result(C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
    capi.check(lib.ramd_prof_spmv_enable(0))
    return cnt.value, avg.value, mn.value, mx.value


for rep in range(2):
    print("back-to-back Apply      ", prof(lambda: A.Apply(x, y), 100))
    print("back-to-back Apply+dot  ", prof(lambda: capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 0)), 100))

    def mix():
        capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 0))
        z.AddScale(w, 0.5)       # 3 vector passes
        w.ScaleAdd(0.5, z)       # 3 vector passes
    print("Apply+dot | 2 axpy-like ", prof(mix, 100))

    def mix2():
        capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 0))
        z.Dot(w)                 # read-only passes + host sync
    print("Apply+dot | dot+sync    ", prof(mix2, 100))
    rhs = ra.LocalVector(); rhs.Allocate("", n); A.Apply(x, rhs)
    ls = S.CG(); ls.SetOperator(A); ls.SetPreconditioner(S.Jacobi()); ls.Init(0.0, 0.0, 1e300, 100); ls.Build()
    sol = ra.LocalVector(); sol.Allocate("", n)
    print("inside CG+Jacobi        ", prof(lambda: (sol.Zeros(), ls.Solve(rhs, sol)), 1))
    ls.Clear()
