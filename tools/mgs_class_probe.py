"""k_mgs_block<4,4> (update with four basis vectors, dots against the next four, one pass over w) timed on operands chosen by
the placement class of their blocks (csrc/backend.hip): which combinations stream well?   python tools/mgs_class_probe.py [nvec]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
ra.init_rocalution()
n = 1 << 27
nvec = int(sys.argv[1]) if len(sys.argv) > 1 else 48
vecs, cls = [], []
for i in range(nvec):
    v = ra.LocalVector(); v.Allocate("v", n); v.Zeros()
    c = C.c_int(-9); capi.check(lib.ramd_vec_placement_class(v._h, C.byref(c)))
    vecs.append(v); cls.append(c.value)
print("classes in allocation order:", "".join(str(c) for c in cls), flush=True)
A = [i for i, c in enumerate(cls) if c == 0]
B = [i for i, c in enumerate(cls) if c == 1]
VT = capi.vec_t * 4


def run(prev, cur, w, reps=6):
    p = VT(*[vecs[i]._h for i in prev]); q = VT(*[vecs[i]._h for i in cur])
    for _ in range(2):
        capi.check(lib.ramd_fused_mgs_block(vecs[w]._h, p, 4, 40, 60, q, 4, 80))
    ra.sync()
    capi.check(lib.ramd_timer_start())
    for _ in range(reps):
        capi.check(lib.ramd_fused_mgs_block(vecs[w]._h, p, 4, 40, 60, q, 4, 80))
    ms = C.c_double(0); capi.check(lib.ramd_timer_stop(C.byref(ms)))
    return ms.value / reps


def pick(pool, k, skip=()):
    out = [i for i in pool if i not in skip][:k]
    return out if len(out) == k else None


rows = []
for name, P, Q, W in (("AAAA|AAAA w A", A, A, A), ("AAAA|AAAA w B", A, A, B), ("AAAA|BBBB w A", A, B, A), ("AAAA|BBBB w B", A, B, B),
                      ("BBBB|BBBB w B", B, B, B), ("BBBB|BBBB w A", B, B, A), ("BBBB|AAAA w A", B, A, A), ("BBBB|AAAA w B", B, A, B)):
    prev = pick(P, 4)
    cur = pick(Q, 4, prev or ())
    w = pick(W, 1, (prev or []) + (cur or []))
    if not (prev and cur and w):
        print(name, ": not enough blocks of a class"); continue
    t = [run(prev, cur, w[0]) for _ in range(2)]
    print("%s: %.3f / %.3f ms  (%.2f TB/s)  blocks %s | %s | %d" % (name, t[0], t[1], 10 * n * 8 / min(t) / 1e9, prev, cur, w[0]), flush=True)
# consecutive blocks as GMRES uses them (allocation order), several offsets
for o in range(0, nvec - 9, 5):
    prev, cur, w = list(range(o, o + 4)), list(range(o + 4, o + 8)), o + 8
    t = run(prev, cur, w)
    print("consecutive from %2d (classes %s|%s w %d): %.3f ms (%.2f TB/s)" % (o, "".join(str(cls[i]) for i in prev), "".join(str(cls[i]) for i in cur), cls[w], t, 10 * n * 8 / t / 1e9), flush=True)
