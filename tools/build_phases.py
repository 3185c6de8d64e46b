"""Where does GMRES+ILU(0)::Build go at N^3?  RAMD_BUILD_VERBOSE=1 python tools/build_phases.py [N]   (GPU box)"""
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
for rep in range(3):
    ls = S.GMRES(); ls.SetOperator(A); ls.SetPreconditioner(S.ILU())
    ra.sync(); t = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t
    print("GMRES+ILU(0) build %.3f s" % tb, file=sys.stderr, flush=True)
    ls.Clear(); ra.sync()
