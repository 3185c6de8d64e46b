"""Do the three kernels of a CG iteration slow each other down?  Each kind alone in a tight loop, then interleaved as the
loop runs them, same vectors, HIP-event totals.  python tools/cg_interplay.py [N]   (GPU box; RAMD_CSR_PAT2 etc. apply)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ra.init_rocalution()
A = ra.LocalMatrix(); A.GenPoisson7(N)
n = N ** 3
vs = {}
for name in ("x", "r", "z", "p", "q", "dinv"):
    v = ra.LocalVector(); v.Allocate(name, n); v.Ones(); vs[name] = v
H = {k: v._h for k, v in vs.items()}


def spmv():
    capi.check(lib.ramd_fused_apply_dot(A._h, H["p"], H["q"], 0))


def update():
    capi.check(lib.ramd_fused_cg_update(H["r"], H["q"], H["dinv"], H["z"], 1, 0, 2, 3))


def direction():
    capi.check(lib.ramd_fused_cg_direction(H["x"], H["p"], H["z"], 1, 0, 3))


def timed(f, reps):
    for _ in range(3):
        f()
    ra.sync()
    capi.check(lib.ramd_timer_start())
    for _ in range(reps):
        f()
    ms = C.c_double(0)
    capi.check(lib.ramd_timer_stop(C.byref(ms)))
    return ms.value / reps


R = 100
a, b, c = timed(spmv, R), timed(update, R), timed(direction, R)


def it():
    update(); direction(); spmv()


d = timed(it, R)
print("PAT2=%s alone: spmv+dot %.3f ms, update %.3f ms, direction %.3f ms (sum %.3f) | interleaved iteration %.3f ms | ratio %.3f"
      % (os.environ.get("RAMD_CSR_PAT2", "0"), a, b, c, a + b + c, d, d / (a + b + c)))
