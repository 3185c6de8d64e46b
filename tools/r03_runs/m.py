import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_cpu_host import _spawn
import _dist_worker as W
from oracle import oracle
oracle.build(); oracle.set_threads(1)
kind = "gr3030"
rp, ci, va = W._matrix(kind)
n = len(rp) - 1
x = np.random.default_rng(5).uniform(-1, 1, n)
yref = oracle.csr_apply(rp, ci, va, x)
res = _spawn("gpu", kind, world=2, timeout=600, env={"RAMD_COMM_HALO": "allgather", "RAMD_COMM_DEBUG": "1"})
y = np.concatenate([r["y"] for r in res]); d = np.abs(y - yref)
print("y bad rows", np.nonzero(d > 1e-12)[0])
y2 = np.concatenate([r["y_ell"] for r in res]); d2 = np.abs(y2 - yref)
print("y_ell bad rows", np.nonzero(d2 > 1e-12)[0])
