"""Does the streaming speed of a kernel depend on WHICH allocations it touches (physical placement), inside one process?
Per-vector read speed, then the 3R+2W CG update on different groups of vectors, each timed with HIP events (tools only)."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import rocalution_amd as ra
from rocalution_amd import capi

lib = capi.load()
ra.init_rocalution()
n = 1 << 27
NV = int(sys.argv[1]) if len(sys.argv) > 1 else 14
vs = []
for k in range(NV):
    v = ra.LocalVector(np.float64); v.Allocate("", n); v.Ones(); vs.append(v)


addr = [lib.ramd_vec_data(v._h) for v in vs]
cls = []
for v in vs:
    c = C.c_int(-9); capi.check(lib.ramd_vec_placement_class(v._h, C.byref(c))); cls.append(c.value)
print("placement classes:", cls, flush=True)
print("addresses:", " ".join("%x" % a for a in addr))
print("offsets to vector 0 in MiB:", " ".join("%.3f" % ((a - addr[0]) / 2**20) for a in addr), flush=True)


def timed(fn, reps=5):
    fn()
    capi.check(lib.ramd_timer_start())
    for _ in range(reps):
        fn()
    ms = C.c_double()
    capi.check(lib.ramd_timer_stop(C.byref(ms)))
    return ms.value / reps


for rep in range(1):
    row = []
    for v in vs:
        h = (capi.vec_t * 1)(v._h)
        ms = timed(lambda: capi.check(lib.ramd_fused_multi_dot(h, 1, v._h, 0)))
        row.append(n * 8 / ms / 1e6)
    print("read GB/s per vector:", " ".join("%5.0f" % x for x in row), flush=True)
groups = [(0, 1, 2, 3), (4, 5, 6, 7), (8, 9, 10, 11), (0, 4, 8, 12), (1, 5, 9, 13), (3, 2, 1, 0), (12, 13, 10, 11), (0, 2, 4, 6), (0, 1, 4, 5), (0, 3, 6, 9), (2, 3, 4, 5), (6, 7, 8, 9), (0, 0, 1, 1)]
for rep in range(1):
    out = []
    for g in groups:
        if max(g) >= NV:
            continue
        r, q, d, z = (vs[i] for i in g)
        capi.check(lib.ramd_scalars_set(1, 1.0)); capi.check(lib.ramd_scalars_set(2, 1e30))
        ms = timed(lambda: capi.check(lib.ramd_fused_cg_update(r._h, q._h, d._h, z._h, 1, 2, 3, 4)))
        out.append("%s[r%d z%d] %.0f" % (g, cls[g[0]], cls[g[3]], 5 * n * 8 / ms / 1e6))
    print("cg_update GB/s:", " | ".join(out), flush=True)
    vals = [float(o.split()[-1]) for o in out[:-1]]
    same = [float(o.split()[-1]) for o, g in zip(out[:-1], groups) if cls[g[0]] == cls[g[3]]]
    diff = [float(o.split()[-1]) for o, g in zip(out[:-1], groups) if cls[g[0]] != cls[g[3]]]
    print("r and z in the SAME class: %s | in DIFFERENT classes: %s" % (" ".join("%.0f" % v for v in same), " ".join("%.0f" % v for v in diff)), flush=True)
    print("cg_update min %.0f max %.0f mean %.0f  (arena phase %s MiB mod %s MiB, arena %s)" % (min(vals), max(vals), sum(vals) / len(vals),
          __import__("os").environ.get("RAMD_ARENA_PHASE_MB", "32"), __import__("os").environ.get("RAMD_ARENA_MOD_MB", "512"),
          __import__("os").environ.get("RAMD_ALLOC_ARENA", "1")), flush=True)
