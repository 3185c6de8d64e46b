"""one rank's share of the 8-way row split of the 512^3 operator (a 512 x 512 x 64 slab: the interior block a rank of the
GlobalMatrix holds), solved on its own: what an iteration costs at that size against an eighth of the full-size iteration"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rocalution_amd as ra
from rocalution_amd import capi, solvers as S
lib = capi.load()
ra.init_rocalution()
N, planes = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = planes * N * N
only = os.environ.get("RAMD_SLAB_ONLY", "")
for name, mk, pc, K in (("cg+jacobi", S.CG, S.Jacobi, 300), ("gmres(30)+ilu(0)", S.GMRES, S.ILU, 90), ("bicgstab+mcsgs", S.BiCGStab, S.MultiColoredSGS, 60)):
    if only and only not in name:
        continue
    A, G = ra.LocalMatrix(), ra.LocalMatrix()
    capi.check(lib.ramd_mat_gen_poisson7_slab(A._h, G._h, N, 0, n))
    del G
    b = ra.LocalVector(); b.Allocate("", n)
    one = ra.LocalVector(); one.Allocate("", n); one.Ones()
    A.Apply(one, b)
    x = ra.LocalVector(); x.Allocate("", n); x.Zeros()
    ls = mk(); ls.SetOperator(A); p = pc(); ls.SetPreconditioner(p)
    if mk is S.GMRES:
        ls.SetBasisSize(30)
    ls.Init(0.0, 0.0, 1e300, K)
    ra.sync(); t0 = time.perf_counter(); ls.Build(); ra.sync(); tb = time.perf_counter() - t0
    for rep in range(2):
        x.Zeros(); ra.sync(); t0 = time.perf_counter(); ls.Solve(b, x); ra.sync(); dt = time.perf_counter() - t0
    it = ls.GetIterationCount()
    print("%s slab 512x512x%d: %d iterations %.3f ms each = %.1f it/s, Build %.2f s" % (name, planes, it, 1e3 * dt / it, it / dt, tb), flush=True)
    ls.Clear()
