"""repeated Build() [+ Solve()] of direct-solve ILU / IC at N^3, to catch outliers:
python tools/build_repeat.py [N] [reps] [solve_iters]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import solvers as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ra.init_rocalution()
A = ra.LocalMatrix(); A.GenPoisson7(N)
n = A.GetM()
ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
rhs = ra.LocalVector(); rhs.Allocate("", n)
x = ra.LocalVector(); x.Allocate("", n)
A.Apply(ones, rhs)
ra.sync()
for rep in range(R):
    for sname, pname in (("GMRES", "ILU"), ("CG", "IC"), ("CG", "SGS")):
        ls = getattr(S, sname)(); ls.SetOperator(A)
        ls.SetPreconditioner(getattr(S, pname)())
        ls.Init(0.0, 0.0, 1e300, max(IT, 1))
        ra.sync(); t = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t
        ts = 0.0
        if IT:
            x.Zeros(); ra.sync(); t = time.perf_counter(); ls.Solve(rhs, x); ra.sync(); ts = time.perf_counter() - t
        t = time.perf_counter(); ls.Clear(); ra.sync(); tc = time.perf_counter() - t
        print("%-10s %-16s build %.3f s  solve %.3f s  clear %.3f s" % (sname, pname, tb, ts, tc), flush=True)
