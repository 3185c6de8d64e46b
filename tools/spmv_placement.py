"""does the CSR SpMV time depend on where x and y live?  14 candidate vectors, every one as x (y fixed) and as y (x fixed)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
ra.init_rocalution()
N = 512
A = ra.LocalMatrix(); A.GenPoisson7(N)
n = N ** 3
vs = []
for k in range(12):
    v = ra.LocalVector(); v.Allocate("", n); v.Ones(); vs.append(v)
cls = []
for v in vs:
    c = C.c_int(-9); capi.check(lib.ramd_vec_placement_class(v._h, C.byref(c))); cls.append(c.value)
print("classes:", cls, flush=True)
def t(x, y, dot, reps=12):
    f = (lambda: capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 5))) if dot else (lambda: A.Apply(x, y))
    f(); f(); ra.sync()
    capi.check(lib.ramd_prof_enable(0, 1))
    for _ in range(reps): f()
    ra.sync()
    cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
    capi.check(lib.ramd_prof_result(0, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
    capi.check(lib.ramd_prof_enable(0, 0))
    return avg.value
for dot in (1,):
    print("x = v_k, y = v_0   :", " ".join("%.3f[%d]" % (t(vs[k], vs[0], dot), cls[k]) for k in range(1, 12)), flush=True)
    print("x = v_0, y = v_k   :", " ".join("%.3f[%d]" % (t(vs[0], vs[k], dot), cls[k]) for k in range(1, 12)), flush=True)
    print("x = v_k, y = v_k+1 :", " ".join("%.3f[%d%d]" % (t(vs[k], vs[k + 1], dot), cls[k], cls[k + 1]) for k in range(1, 11)), flush=True)
