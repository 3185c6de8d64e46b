"""triangular solves under a SHARED device: LUSolve on Poisson N^3 in a chosen form against the CPU oracle, repeated; run several
copies at once (tools/_jobs) to see whether another process on the GPU changes a result.
    python tools/lat_stress.py N reps tag [lat|ct|level|sf]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import generators as gen
from oracle import oracle
oracle.build(); oracle.set_threads(1)
ra.init_rocalution()
N, reps, tag = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
form = sys.argv[4] if len(sys.argv) > 4 else "lat"
n = N ** 3
rp, ci, va = gen.poisson7(N)
lu = oracle.ilu0(rp, ci, va)
A = ra.LocalMatrix(); A.GenPoisson7(N); A.ILU0Factorize()
env = {"lat": {}, "ct": {"RAMD_TRSV_LAT": "0", "RAMD_TRSV_CT_MINROWS": "0", "RAMD_TRSV_CT_MINLEN": "0"},
       "level": {"RAMD_TRSV_LAT": "0", "RAMD_TRSV_CT": "0"}, "sf": {"RAMD_TRSV_LAT": "0", "RAMD_TRSV_CT": "0", "RAMD_TRSV_SF": "2"}}[form]
os.environ.update(env)
A.LUAnalyse()
rng = np.random.default_rng(1)
y = ra.LocalVector(); y.Allocate("", n)
bad = 0
t0 = time.time()
for rep in range(reps):
    bh = rng.uniform(-1, 1, n)
    want = oracle.lusolve(rp, ci, lu, bh)
    b = ra.LocalVector(data=bh)
    for k in range(4):
        A.LUSolve(b, y)
        got = y.numpy()
        if not np.array_equal(got, want):
            idx = np.flatnonzero(got != want)
            bad += 1
            if bad <= 3:
                print(tag, form, "rep", rep, k, "MISMATCH", len(idx), "rows; first", [(int(i % N), int((i // N) % N), int(i // (N * N))) for i in idx[:6]],
                      "max diff", float(np.max(np.abs(got - want))), flush=True)
print(tag, form, "N", N, "reps", reps, "mismatching solves", bad, "of", reps * 4, "in %.1f s" % (time.time() - t0), flush=True)
