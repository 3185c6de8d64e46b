"""one process: ILU(0) build at N^3 (twice), preconditioner apply, print times + checksums (hunting a rare slow k_ilu0)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import solvers as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ra.init_rocalution()
A = ra.LocalMatrix(); A.GenPoisson7(N)
n = A.GetM()
ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
z = ra.LocalVector(); z.Allocate("", n)
for rep in range(2):
    ls = S.GMRES(); ls.SetOperator(A); ls.SetPreconditioner(S.ILU())
    ra.sync(); t = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t
    ls.PrecondApply(ones, z)
    zz = z.numpy()
    print("build %.3f s  sum %.17g  norm %.17g  nan %d" % (tb, zz.sum(), np.linalg.norm(zz), int(np.isnan(zz).sum())), flush=True)
    ls.Clear()
