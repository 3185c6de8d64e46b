"""small staged probe of the sync-free kernels (run under `timeout`)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import generators as gen
from oracle import oracle as orc
orc.build(); orc.set_threads(1)
ra.init_rocalution()
for N in (4, 8, 20, 48):
    rp, ci, va = gen.poisson7(N)
    n = len(rp) - 1
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    t0 = time.time(); A.ILU0Factorize(); ra.sync(); t1 = time.time()
    lu = orc.ilu0(rp, ci, va)
    print("N=%d ilu0 %.4fs exact=%s" % (N, t1 - t0, np.array_equal(A.CopyToCSR()[2], lu)), flush=True)
    t0 = time.time(); A.LUAnalyse(); ra.sync(); t1 = time.time()
    print("   analyse %.4fs" % (t1 - t0), flush=True)
    b = np.random.default_rng(N).uniform(-1, 1, n)
    y = ra.LocalVector(); y.Allocate("", n)
    vb = ra.LocalVector(data=b)
    t0 = time.time(); A.LUSolve(vb, y); ra.sync(); t1 = time.time()
    print("   lusolve %.4fs exact=%s" % (t1 - t0, np.array_equal(y.numpy(), orc.lusolve(rp, ci, lu, b))), flush=True)
print("DONE")
