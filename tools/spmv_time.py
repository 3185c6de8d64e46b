"""SpMV timing (HIP events of the SPMV channel):  python tools/spmv_time.py [N] [reps] [format] [poisson|lap27]   (env knobs of spmv.hip apply)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
fmt = sys.argv[3] if len(sys.argv) > 3 else "csr"
ra.init_rocalution()
which = sys.argv[4] if len(sys.argv) > 4 else "poisson"
A = ra.LocalMatrix()
if which == "lap27":
    A.GenLaplace27(N)
else:
    A.GenPoisson7(N)
n = N ** 3; nnz = A.GetNnz()
x = ra.LocalVector(); x.Allocate("x", n); x.Ones()
y = ra.LocalVector(); y.Allocate("y", n)
if fmt != "csr":
    A.ConvertTo({"ell": ra.ELL, "hyb": ra.HYB}[fmt])
dot = os.environ.get("DOT", "0") == "1"  # the fused Apply + <x, y> form the CG loop uses
def once():
    if dot:
        capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 5))
    else:
        A.Apply(x, y)
for _ in range(5):
    once()
ra.sync()
capi.check(lib.ramd_prof_enable(0, 1))
for _ in range(reps):
    once()
ra.sync()
cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
capi.check(lib.ramd_prof_result(0, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
B = 4 * (n + nnz) + 8 * (2 * n + nnz) if fmt == "csr" else (4 + 8) * (27 if which == "lap27" else 7) * n + 16 * n
print(("spmv+dot" if dot else "spmv") + " %s %d^3: avg %.4f ms (min %.4f max %.4f, %d launches) = %.1f GB/s algorithmic = %.3f of 8 TB/s | norm %.17g | tag=%s"
      % (fmt, N, avg.value, mn.value, mx.value, cnt.value, B / avg.value / 1e6, B / avg.value / 1e6 / 8000, y.Norm(), os.environ.get("TAG", "")), flush=True)
