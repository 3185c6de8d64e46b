"""CSR SpMV on the config-3 surrogate: HIP-event average of Apply and of the fused Apply + dot; env knobs of spmv.hip apply"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi, generators as gen
lib = capi.load()
ra.init_rocalution()
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 549
rp, ci, va = gen.shell_surrogate(nx)
n, nnz = len(rp) - 1, len(ci)
A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
x = ra.LocalVector(data=np.random.default_rng(1).uniform(-1, 1, n)); y = ra.LocalVector(); y.Allocate("y", n)
for dot in (0, 1):
    f = (lambda: capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 5))) if dot else (lambda: A.Apply(x, y))
    for _ in range(5):
        f()
    ra.sync()
    capi.check(lib.ramd_prof_enable(0, 1))
    for _ in range(200):
        f()
    ra.sync()
    cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
    capi.check(lib.ramd_prof_result(0, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
    capi.check(lib.ramd_prof_enable(0, 0))
    B = 4 * (n + nnz) + 8 * (2 * n + nnz)
    st = C.c_int(0); capi.check(lib.ramd_mat_pattern_info(A._h, C.byref(st), None, None))
    print("shell %d spmv%s: avg %.4f ms (min %.4f) = %.0f GB/s algorithmic = %.3f of 8 TB/s | state %d | checksum %.17g | tag=%s"
          % (nx, "+dot" if dot else "", avg.value, mn.value, B / avg.value / 1e6, B / avg.value / 8e9, st.value, float(np.sum(y.numpy())), os.environ.get("TAG", "")), flush=True)
