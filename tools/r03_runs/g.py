import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import solvers as S
from test_gpu_solvers import _mk, load_golden
ra.init_rocalution()
for name in ("gr3030", "poisson8", "rand300"):
    g = load_golden(name)
    A = ra.LocalMatrix(); A.SetDataPtrCSR(g["rowptr"], g["col"], g["val"])
    n = A.GetM()
    x = ra.LocalVector(data=g["x"])
    for tag, key in (("cg_jacobi", "pc_jacobi"), ("cg_ilu0", "pc_ilu0"), ("cg_ilu1", "pc_ilu1"), ("cg_mcsgs", "pc_mcsgs"), ("cg_gs", "pc_gs"),
                     ("cg_sgs", "pc_sgs"), ("cg_ic", "pc_ic")):
        if key not in g:
            continue
        sys.stderr.write("--- %s %s\n" % (name, tag)); sys.stderr.flush()
        ls = _mk(S, tag); ls.SetOperator(A); ls.Build()
        z = ra.LocalVector(); z.Allocate("", n)
        ls.PrecondApply(x, z)
        d = np.max(np.abs(z.numpy() - g[key]))
        print(name, tag, "maxdiff", d, flush=True)
