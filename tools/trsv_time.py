"""time the sparse triangular solves of ILU(0) (LUSolve) on the config-3 surrogate or a Poisson grid; env knobs of trisolve.hip apply
    python tools/trsv_time.py shell 549 | poisson 512"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi, generators as gen
lib = capi.load()
ra.init_rocalution()
kind, N = sys.argv[1], int(sys.argv[2])
A = ra.LocalMatrix()
if kind == "shell":
    rp, ci, va = gen.shell_surrogate(N)
    A.SetDataPtrCSR(rp, ci, va)
else:
    A.GenPoisson7(N)
n = A.GetM()
t0 = time.time(); A.ILU0Factorize(); ra.sync(); t1 = time.time(); A.LUAnalyse(); ra.sync(); t2 = time.time()
b = ra.LocalVector(); b.Allocate("b", n); b.Ones()
y = ra.LocalVector(); y.Allocate("y", n)
for _ in range(5):
    A.LUSolve(b, y)
ra.sync()
capi.check(lib.ramd_prof_enable(1, 1))
R = 40
t3 = time.time()
for _ in range(R):
    A.LUSolve(b, y)
ra.sync(); t4 = time.time()
cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
capi.check(lib.ramd_prof_result(1, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
print("%s %d: ilu0 %.3fs analyse %.3fs | LUSolve %.3f ms wall; per triangle avg %.3f min %.3f max %.3f ms (%d launches) | tag=%s"
      % (kind, N, t1 - t0, t2 - t1, (t4 - t3) / R * 1e3, avg.value, mn.value, mx.value, cnt.value, os.environ.get("TAG", "")), flush=True)
