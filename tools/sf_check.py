"""the sync-free grouped triangular solve (k_trsv_sf) on the config-3 class in one numbering: bit-exact against the other forms
(small size: the oracle; full size: the level-scheduled kernel of a second matrix object), then timed
    python tools/sf_check.py rcm 549 [reps]      (env knobs of trisolve.hip apply)"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi, generators as gen
lib = capi.load()
ra.init_rocalution()
kind, N = sys.argv[1], int(sys.argv[2])
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rp, ci, va = gen.shell_variant(N, kind) if kind != "lex" else gen.shell_surrogate(N)
n = len(rp) - 1
A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
t0 = time.time(); A.ILU0Factorize(); ra.sync(); t1 = time.time(); A.LUAnalyse(); ra.sync(); t2 = time.time()
st = (C.c_int64 * 16)()
forms = []
for which in (0, 1):
    capi.check(lib.ramd_tri_plan_stats(which, st)); forms.append(list(st))
bh = np.random.default_rng(7).uniform(-1, 1, n)
b = ra.LocalVector(data=bh)
y = ra.LocalVector(); y.Allocate("y", n)
for _ in range(3):
    A.LUSolve(b, y)
ra.sync()
got = y.numpy().copy()
if os.environ.get("SF_REF"):
    ref = np.load(os.environ["SF_REF"])
    print("bit-exact vs reference file:", bool(np.array_equal(ref, got)), "max |diff|", float(np.max(np.abs(ref - got))), flush=True)
if os.environ.get("SF_SAVE"):
    np.save(os.environ["SF_SAVE"], got)
capi.check(lib.ramd_prof_enable(1, 1))
t3 = time.time()
for _ in range(R):
    A.LUSolve(b, y)
ra.sync(); t4 = time.time()
cnt, avg, mn, mx = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
capi.check(lib.ramd_prof_result(1, C.byref(cnt), C.byref(avg), C.byref(mn), C.byref(mx)))
same = bool(np.array_equal(got, y.numpy()))
print("%s %d n=%d: ilu0 %.3fs analyse %.3fs | forms L %s U %s | LUSolve %.3f ms wall; per triangle avg %.3f min %.3f max %.3f ms (%d launches) | repeatable %s | tag=%s"
      % (kind, N, n, t1 - t0, t2 - t1, forms[0][:9], forms[1][:9], (t4 - t3) / R * 1e3, avg.value, mn.value, mx.value, cnt.value, same,
         os.environ.get("TAG", "")), flush=True)
