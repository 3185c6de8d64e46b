"""ILU(p) on the 3-D Poisson operator: factorisation + analysis time and GMRES(30) rate.  python tools/ilup_time.py N p"""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra
from rocalution_amd import solvers as S
N = int(sys.argv[1]); p = int(sys.argv[2]); level = (len(sys.argv) < 4 or sys.argv[3] != "0")
ra.init_rocalution()
A = ra.LocalMatrix(); A.GenPoisson7(N)
n = A.GetM()
ones = ra.LocalVector(data=np.ones(n)); rhs = ra.LocalVector(); rhs.Allocate("", n); A.Apply(ones, rhs)
for rep in range(int(os.environ.get("REPS", "2"))):
    ls = S.GMRES(); pc = S.ILU(); pc.Set(p, level); ls.SetPreconditioner(pc); ls.SetOperator(A)
    ls.Init(1e-15, 1e-8, 1e8, 300)
    t0 = time.time(); ls.Build(); ra.sync(); tb = time.time() - t0
    x = ra.LocalVector(); x.Allocate("", n)
    t0 = time.time(); ls.Solve(rhs, x); ra.sync(); ts = time.time() - t0
    it = ls.GetIterationCount()
    err = np.abs(x.numpy() - 1).max()
    print("N=%d p=%d level=%d build %.3f s  solve %.3f s  iters %d (%.1f it/s) status %d  max|x-1| %.2e"
          % (N, p, level, tb, ts, it, it / ts, ls.GetSolverStatus(), err), flush=True)
    ls.Clear()
