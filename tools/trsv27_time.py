"""Triangular-solve timing on the 27-point operator's ILU(0) factors (HIP events of the TRSV channel: one launch per triangle):
python tools/trsv27_time.py [N] [reps]   (env knobs of trsv_box27.hip / trisolve.hip apply)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
CH = 1  # RAMD_PROF_TRSV
ra.init_rocalution()
A = ra.LocalMatrix(); A.GenLaplace27(N)
A.ILU0Factorize(); A.LUAnalyse()
n = N ** 3
b = ra.LocalVector(); b.Allocate("", n); b.Ones()
y = ra.LocalVector(); y.Allocate("", n)
for _ in range(3):
    A.LUSolve(b, y)
ra.sync()
capi.check(lib.ramd_prof_enable(CH, 1))
for _ in range(reps):
    A.LUSolve(b, y)
ra.sync()
cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
capi.check(lib.ramd_prof_result(CH, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
nnz_tri = (A.GetNnz() - n) // 2
Bt = 12 * nnz_tri + 8 * n + 4 * n + 16 * n  # one triangle in CSR terms: entries, diagonal, row pointers, in and out
print("trsv lap27 %d^3: avg %.4f ms per triangle (min %.4f max %.4f, %d launches) = %.3f of 8 TB/s on the CSR bytes | norm %.17g | tag=%s"
      % (N, avg.value, mn.value, mx.value, cnt.value, Bt / avg.value / 1e6 / 8000, y.Norm(), os.environ.get("TAG", "")), flush=True)
