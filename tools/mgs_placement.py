"""Does the full Gram-Schmidt block pass (w updated by 4 vectors, projected on 4 more: 9 reads + 1 write) care which
placement classes its ten vectors are in?  python tools/mgs_placement.py   (GPU box)"""
import ctypes as C, os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi
lib = capi.load()
ra.init_rocalution()
n = 512 ** 3
NV = 26
vs = []
for i in range(NV):
    v = ra.LocalVector(); v.Allocate("v%d" % i, n); v.Ones(); vs.append(v)
cls = []
for v in vs:
    c = C.c_int(-1); capi.check(lib.ramd_vec_placement_class(v._h, C.byref(c))); cls.append(c.value)
print("classes", cls, flush=True)
K = lib.ramd_fused_mgs_block_max()
nsum = K + K * (K - 1) // 2
area1 = 512 - 4 - nsum; area0 = area1 - nsum


def run(w, vp, vc, reps=20):
    hp = (capi.vec_t * K)(*[v._h for v in vp]); hc = (capi.vec_t * K)(*[v._h for v in vc])
    for _ in range(3):
        capi.check(lib.ramd_fused_mgs_block(w._h, hp, K, 0, area0, hc, K, area1))
    ra.sync(); capi.check(lib.ramd_timer_start())
    for _ in range(reps):
        capi.check(lib.ramd_fused_mgs_block(w._h, hp, K, 0, area0, hc, K, area1))
    ms = C.c_double(0); capi.check(lib.ramd_timer_stop(C.byref(ms)))
    return ms.value / reps


c0 = [i for i in range(NV) if cls[i] == 0]; c1 = [i for i in range(NV) if cls[i] == 1]
print("class 0:", len(c0), "class 1:", len(c1), flush=True)
big, small = (c0, c1) if len(c0) >= len(c1) else (c1, c0)
if len(big) >= 9:
    s = big[:9]
    print("all nine in one class: %.3f ms" % run(vs[s[0]], [vs[i] for i in s[1:5]], [vs[i] for i in s[5:9]]), flush=True)
    s = big[9:18] if len(big) >= 18 else big[:9][::-1]
    print("all nine in one class (other blocks): %.3f ms" % run(vs[s[0]], [vs[i] for i in s[1:5]], [vs[i] for i in s[5:9]]), flush=True)
if len(small) >= 1 and len(big) >= 8:
    print("w in the other class, eight reads in one: %.3f ms" % run(vs[small[0]], [vs[i] for i in big[0:4]], [vs[i] for i in big[4:8]]), flush=True)
if len(small) >= 4 and len(big) >= 5:
    print("mixed (4 reads other class): %.3f ms" % run(vs[big[0]], [vs[i] for i in small[0:4]], [vs[i] for i in big[1:5]]), flush=True)
    print("mixed (alternating): %.3f ms" % run(vs[big[0]], [vs[small[0]], vs[big[1]], vs[small[1]], vs[big[2]]],
                                              [vs[small[2]], vs[big[3]], vs[small[3]], vs[big[4]]]), flush=True)
for k in range(3):
    idx = list(np.random.default_rng(k).permutation(NV)[:9])
    print("random nine %s (classes %s): %.3f ms" % (idx, [cls[i] for i in idx], run(vs[idx[0]], [vs[i] for i in idx[1:5]], [vs[i] for i in idx[5:9]])), flush=True)
