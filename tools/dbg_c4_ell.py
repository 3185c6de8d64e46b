import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_global_full_size as T
body = r'''
for fmt in (ra.ELL, ra.HYB, ra.CSR):
    g = C.c_void_p()
    capi.check(lib.ramd_gsolver_create(comm, capi.SOLVER_BICGSTAB, capi.PC_MCSGS, C.byref(g)))
    capi.check(lib.ramd_gsolver_init(g, 1e-15, 1e-6, 1e8, 0, 5000))
    xg, itg, stg, rsg, nred = gsolve(g, fmt)
    print("fmt", fmt, "iters", itg, "status", stg, "res", rsg, "rms err", np.sqrt(np.mean((xg - 1.0) ** 2)), "max err", np.max(np.abs(xg - 1.0)), flush=True)
    capi.check(lib.ramd_gsolver_destroy(g))
'''
exec(T._PRELUDE + body)
