"""where does MultiColoredSGS::Build spend its time? (GPU box)  python tools/build_profile.py [N]"""
import ctypes as C
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import capi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ra.init_rocalution()
lib = capi.load()
A = ra.LocalMatrix(); A.GenPoisson7(N)
ra.sync()


def timed(name, f):
    ra.sync(); t = time.perf_counter(); r = f(); ra.sync()
    print("%-28s %8.3f s" % (name, time.perf_counter() - t), flush=True)
    return r


for rep in range(2):
    B = ra.LocalMatrix()
    timed("clone", lambda: B.CloneFrom(A))
    nc, sizes, perm = timed("multicoloring", lambda: A.MultiColoring())
    timed("permute", lambda: B.Permute(perm))
    h = capi.ptr() if hasattr(capi, "ptr") else C.c_void_p()
    sz = np.ascontiguousarray(sizes, dtype=np.int32)
    timed("mcsgs_build (sweep plans)", lambda: capi.check(lib.ramd_mcsgs_build(
        B._h, nc, sz.ctypes.data_as(C.POINTER(C.c_int32)), perm._h, C.byref(h))))
    lib.ramd_mcsgs_destroy(h)
    C2 = ra.LocalMatrix(); C2.CloneFrom(A)
    timed("ilu0 factorize", lambda: C2.ILU0Factorize())
    timed("lu analyse", lambda: C2.LUAnalyse())
    print("colors", nc, sizes)
