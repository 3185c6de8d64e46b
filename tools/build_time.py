"""Build() time of solver+preconditioner pairs at N^3 (GPU box):  python tools/build_time.py [N]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import solvers as S  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ra.init_rocalution()
A = ra.LocalMatrix(); A.GenPoisson7(N)
ra.sync()
for rep in range(2):
    for sname, pname in (("BiCGStab", "MultiColoredGS"), ("BiCGStab", "MultiColoredSGS"), ("BiCGStab", "MultiColoredILU"),
                         ("GMRES", "ILU"), ("CG", "Jacobi"), ("BiCGStab", None)):
        ls = getattr(S, sname)(); ls.SetOperator(A)
        if pname:
            ls.SetPreconditioner(getattr(S, pname)())
        ra.sync(); t = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t
        t = time.perf_counter(); ls.Clear(); ra.sync(); tc = time.perf_counter() - t
        print("%-10s %-16s build %.3f s  clear %.3f s" % (sname, pname, tb, tc), flush=True)
