import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_cpu_host import _spawn
import _dist_worker as W
from oracle import oracle
oracle.build(); oracle.set_threads(1)
for kind in ("random", "gr3030"):
    rp, ci, va = W._matrix(kind)
    if kind == "random":
        rp, ci, va = W._symmetrize_pattern(rp, ci, va)
    n = len(rp) - 1
    x = np.random.default_rng(5).uniform(-1, 1, n)
    yref = oracle.csr_apply(rp, ci, va, x)
    for world in (2, 4):
        for form in ("sendrecv", "allgather"):
            try:
                res = _spawn("gpu", kind, world=world, timeout=600, env={"RAMD_COMM_HALO": form})
                y = np.concatenate([r["y"] for r in res])
                d = np.abs(y - yref)
                print(kind, world, form, "max diff", d.max(), "bad rows", np.nonzero(d > 1e-12)[0][:10], "it", [int(r["it"]) for r in res], flush=True)
            except Exception as e:
                print(kind, world, form, "FAILED", repr(e)[:300], flush=True)
