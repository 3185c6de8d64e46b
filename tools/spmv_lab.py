"""A/B the CSR SpMV kernel variants of csrc/lab.hip (GPU box)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
lib.ramdx_lab_csr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
lib.ramdx_lab_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
ra.init_rocalution()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(9))
A = ra.LocalMatrix(); A.GenPoisson7(N)
n, nnz = A.GetM(), A.GetNnz()
x = ra.LocalVector(); x.Allocate("", n); x.Ones()
y = ra.LocalVector(); y.Allocate("", n)
yref = ra.LocalVector(); yref.Allocate("", n)
A.Apply(x, yref)
ms = C.c_double(0)
B = 4 * (n + nnz) + 8 * (2 * n + nnz)
capi.check(lib.ramdx_lab_copy(x._h, y._h, 0, 20, C.byref(ms)))
print("copy   %8.3f ms %8.1f GB/s" % (ms.value, 16 * n / ms.value / 1e6))
capi.check(lib.ramdx_lab_copy(x._h, y._h, 1, 20, C.byref(ms)))
print("read   %8.3f ms %8.1f GB/s" % (ms.value, 8 * n / ms.value / 1e6))
for v in variants:
    y.Zeros()
    capi.check(lib.ramdx_lab_csr(A._h, x._h, y._h, v, 20, C.byref(ms)))
    y.AddScale(yref, -1.0)
    print("v%-2d    %8.3f ms %8.1f GB/s  (%.1f%% of 8TB/s)  err=%g" % (v, ms.value, B / ms.value / 1e6, B / ms.value / 1e6 / 80, y.Norm() if v != 7 else -1))
