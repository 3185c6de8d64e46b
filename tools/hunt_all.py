"""one process: the sync-free kernels at N^3 (colouring, level analysis, ILU(0)/IC factorisation, level-scheduled and
multicolour sweeps) -- build + short solve per pair, wall times and a checksum (hunting rare stalls)"""
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
rhs = ra.LocalVector(); rhs.Allocate("", n); A.Apply(ones, rhs)
x = ra.LocalVector(); x.Allocate("", n)
for sname, pname, its in (("BiCGStab", "MultiColoredSGS", 10), ("GMRES", "ILU", 20), ("CG", "IC", 20), ("CG", "SGS", 10),
                          ("GMRES", "MultiColoredILU", 20)):
    ls = getattr(S, sname)(); ls.SetOperator(A); ls.SetPreconditioner(getattr(S, pname)())
    ls.Init(0.0, 0.0, 1e300, its)
    ra.sync(); t = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t
    x.Zeros(); ra.sync(); t = time.perf_counter(); ls.Solve(rhs, x); ra.sync(); ts = time.perf_counter() - t
    print("%-9s %-16s build %7.3f s  solve %7.3f s  res %.15e" % (sname, pname, tb, ts, ls.GetCurrentResidual()), flush=True)
    ls.Clear()
