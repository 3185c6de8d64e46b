"""soak of the sync-free grouped triangular solve: N LUSolves on one numbering of the config-3 class, every result compared bit for
bit with the first (a hand-off that lets a value through early shows as a differing solution); run several at once to put the
kernels of different processes side by side on the device (units by ticket: no wave waits for one that is not running)
    python tools/sf_soak.py rcm 549 200"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import generators as gen
ra.init_rocalution()
kind, N, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rp, ci, va = gen.shell_variant(N, kind)
n = len(rp) - 1
A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
A.ILU0Factorize(); A.LUAnalyse()
b = ra.LocalVector(data=np.random.default_rng(11).uniform(-1, 1, n))
y = ra.LocalVector(); y.Allocate("y", n)
A.LUSolve(b, y); ra.sync()
ref = y.numpy().copy()
bad = 0
t0 = time.time()
for i in range(R):
    A.LUSolve(b, y)
    if i % 10 == 9 or i == R - 1:
        if not np.array_equal(y.numpy(), ref):
            bad += 1
ra.sync()
print("%s %d pid %d: %d solves in %.2f s (%.2f ms each), differing checks %d, finite %s" % (kind, N, os.getpid(), R, time.time() - t0, (time.time() - t0) / R * 1e3, bad, bool(np.isfinite(ref).all())), flush=True)
sys.exit(1 if bad else 0)
