"""Where does GMRES+ILU(0)::Build go on the config-3 surrogate?  RAMD_BUILD_VERBOSE=1 python tools/build_phases_shell.py [nx]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import solvers as S  # noqa: E402
from rocalution_amd import generators as gen  # noqa: E402

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 549
ra.init_rocalution()
rp, ci, va = gen.shell_surrogate(nx)
A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
ra.sync()
for rep in range(3):
    ls = S.GMRES(); ls.SetOperator(A); ls.SetPreconditioner(S.ILU())
    ra.sync(); t = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t
    print("GMRES+ILU(0) build %.3f s" % tb, file=sys.stderr, flush=True)
    ls.Clear(); ra.sync()
