"""does the time of the upper solve depend on where its output vector lives?  python tools/trsv_out_placement.py 512"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
ra.init_rocalution()
N = int(sys.argv[1])
A = ra.LocalMatrix(); A.GenPoisson7(N); n = A.GetM()
A.ILU0Factorize(); A.LUAnalyse(); ra.sync()
b = ra.LocalVector(); b.Allocate("b", n); b.Ones()
keep = []
def timed(bb, yy, R=20):
    for _ in range(3):
        A.LUSolve(bb, yy)
    ra.sync()
    capi.check(lib.ramd_prof_enable(1, 1))
    for _ in range(R):
        A.LUSolve(bb, yy)
    ra.sync()
    cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
    capi.check(lib.ramd_prof_result(1, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
    capi.check(lib.ramd_prof_enable(1, 0))
    return avg.value, mn.value, mx.value
for k in range(8):
    y = ra.LocalVector(); y.Allocate("y%d" % k, n); keep.append(y)
    print("output vector %d: per triangle avg %.3f min(L) %.3f max(U) %.3f ms" % ((k,) + timed(b, y)), flush=True)
for k in range(4):
    b2 = ra.LocalVector(); b2.Allocate("b%d" % k, n); b2.Ones(); keep.append(b2)
    print("right-hand side %d (output 0): per triangle avg %.3f min(L) %.3f max(U) %.3f ms" % ((k,) + timed(b2, keep[0])), flush=True)
