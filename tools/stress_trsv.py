"""Soak of the box-tile triangular solve: the same LUSolve many times, every result compared bit for bit with the first
(hand-off races show up as rare mismatches).  python tools/stress_trsv.py [N] [reps]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import generators as gen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ra.init_rocalution()
for name, (rp, ci, va) in (("poisson%d" % N, gen.poisson7(N)), ("shell%d" % (N // 2), gen.shell_surrogate(N // 2, N // 2))):
    n = len(rp) - 1
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    A.ILU0Factorize(); A.LUAnalyse()
    b = ra.LocalVector(data=np.random.default_rng(1).uniform(-1, 1, n))
    y = ra.LocalVector(); y.Allocate("", n)
    A.LUSolve(b, y)
    first = y.numpy().copy()
    assert np.isfinite(first).all()
    t0 = time.time()
    bad = 0
    for r in range(reps):
        A.LUSolve(b, y)
        if r % 10 == 0 or r == reps - 1:
            if not np.array_equal(y.numpy(), first):
                bad += 1
    print("%s: n=%d, %d solves in %.1f s, mismatching samples: %d" % (name, n, reps, time.time() - t0, bad), flush=True)
    assert bad == 0
print("stress ok")
