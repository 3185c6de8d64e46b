"""GPU parity at BASELINE.json's FULL sizes (3-D Poisson 256^3 and 512^3, fp64), where the oracle would take
minutes: size-independent properties of the domain instead of element-wise comparison.
  * A*1 is known in closed form (6 - number of neighbours), exactly representable: bit-exact check of the SpMV
  * CSR, ELL and HYB keep the row order of the entries -> their SpMV results must be bit-identical
  * linearity A(a x + b y) = a A x + b A y to round-off;  <x,Ay> = <Ax,y> (symmetric operator)
  * the fused <x,Ax> equals the separate dot
  * CG+Jacobi converges to the known solution in the iteration count the small-grid oracle runs extrapolate to
    (kappa ~ N^2 -> iterations ~ N), residual history monotone in the A-norm sense (checked via <r,z> > 0)
  * M^-1 = (LU)^-1 of ILU(0): L U (LUSolve(b)) == b on the pattern  (factor * solve round trip)
  * MC-SGS: the fused colour sweeps and the block form agree bit-for-bit at full size
  * iterative triangular solves: a triangular Jacobi iteration is exact after (number of levels) sweeps, so ILU(0)
    with 3N sweeps must reproduce the level-scheduled solve to round-off; the device stopping test must end a run
    with a loose tolerance early (far cheaper) without changing more than that tolerance allows
  * CG+IC reaches the known solution in fewer iterations than CG+Jacobi
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ra():
    import rocalution_amd as ra
    ra.init_rocalution()
    return ra


@pytest.fixture(scope="module")
def S():
    from rocalution_amd import solvers
    return solvers


def _expected_row_sums(N):
    i = np.arange(N)
    nb1 = (i > 0).astype(np.int8) + (i < N - 1).astype(np.int8)  # neighbours along one axis
    nb = nb1[:, None, None] + nb1[None, :, None] + nb1[None, None, :]  # [z, y, x]
    return (6 - nb).astype(np.float64).ravel()


@pytest.mark.parametrize("N", [256, 512])
def test_spmv_closed_form_and_formats_bit_identical(ra, N):
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    assert A.GetNnz() == 7 * n - 6 * N * N
    ones = ra.LocalVector(); ones.Allocate("1", n); ones.Ones()
    y = ra.LocalVector(); y.Allocate("y", n)
    A.Apply(ones, y)
    ref = _expected_row_sums(N)
    got = y.numpy()
    assert np.array_equal(got, ref)
    # a non-trivial x: all three formats must agree bit-for-bit (same per-row order of the products)
    rng = np.random.default_rng(N)
    xh = rng.uniform(-4.0, 6.0, n)
    x = ra.LocalVector(data=xh)
    A.Apply(x, y)
    y_csr = y.numpy().copy()
    for fmt in (ra.ELL, ra.HYB):
        assert A.ConvertTo(fmt) == fmt
        A.Apply(x, y)
        assert np.array_equal(y.numpy(), y_csr)
    # closed form on a sample of rows (7-point stencil evaluated on the host in the same order)
    idx = rng.integers(0, n, 2000)
    N2 = N * N
    for r in idx:
        z_, rem = divmod(int(r), N2); y_, x_ = divmod(rem, N)
        s = 0.0
        if z_ > 0: s += -1.0 * xh[r - N2]
        if y_ > 0: s += -1.0 * xh[r - N]
        if x_ > 0: s += -1.0 * xh[r - 1]
        s += 6.0 * xh[r]
        if x_ < N - 1: s += -1.0 * xh[r + 1]
        if y_ < N - 1: s += -1.0 * xh[r + N]
        if z_ < N - 1: s += -1.0 * xh[r + N2]
        assert y_csr[r] == s


def test_linearity_symmetry_and_fused_dot_512(ra):
    from rocalution_amd import capi
    lib = capi.load()
    N = 512
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    rng = np.random.default_rng(7)
    x = ra.LocalVector(data=rng.uniform(-1, 1, n)); y = ra.LocalVector(data=rng.uniform(-1, 1, n))
    Ax = ra.LocalVector(); Ax.Allocate("", n); Ay = ra.LocalVector(); Ay.Allocate("", n)
    A.Apply(x, Ax); A.Apply(y, Ay)
    # symmetry: <x, A y> == <A x, y>
    a, b = x.Dot(Ay), Ax.Dot(y)
    assert abs(a - b) <= 1e-11 * max(abs(a), abs(b), 1.0)
    # fused <x, A x> == separate dot
    w = ra.LocalVector(); w.Allocate("", n)
    capi.check(lib.ramd_fused_apply_dot(A._h, x._h, w._h, 9))
    out = (C.c_double * 1)()
    capi.check(lib.ramd_scalars_fetch(out, 9, 1))
    assert np.array_equal(w.numpy(), Ax.numpy())
    assert abs(out[0] - x.Dot(Ax)) <= 1e-12 * abs(out[0])
    # linearity: A(2.5 x - 0.75 y) == 2.5 A x - 0.75 A y  (to round-off of the 7-term row sums)
    z = ra.LocalVector(data=x.numpy()); z.ScaleAddScale(2.5, y, -0.75)
    Az = ra.LocalVector(); Az.Allocate("", n)
    A.Apply(z, Az)
    comb = ra.LocalVector(data=Ax.numpy()); comb.ScaleAddScale(2.5, Ay, -0.75)
    comb.AddScale(Az, -1.0)
    assert comb.Norm() <= 1e-13 * Az.Norm() * 10


def test_cg_jacobi_converges_to_known_solution_512(ra, S):
    """the headline workload run to convergence: x -> 1, iteration count in the range the reference's own
    scaling predicts (32^3: 66, 64^3: ~130, ... ~ 2N + small)"""
    N = 512
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
    rhs = ra.LocalVector(); rhs.Allocate("", n)
    A.Apply(ones, rhs)
    ls = S.CG(); ls.SetOperator(A); ls.SetPreconditioner(S.Jacobi()); ls.Init(1e-15, 1e-8, 1e8, 5000); ls.Build()
    x = ra.LocalVector(); x.Allocate("", n)
    ls.Solve(rhs, x)
    it, st = ls.GetIterationCount(), ls.GetSolverStatus()
    assert st == 2 and 900 <= it <= 1700, (it, st)
    h = np.asarray(ls.GetResidualHistory())
    assert h[-1] <= 1e-8 * h[0]
    x.AddScale(ones, -1.0)
    assert x.Norm() / np.sqrt(n) < 1e-6  # rms error


def test_ilu0_factor_solve_round_trip_256(ra, S):
    """ILU(0) of the 7-point operator at 256^3: x = (LU)^-1 b, then L (U x) == b to round-off, with L and U
    applied as SpMV of the factor parts (checks factorisation, level analysis and both sync-free solves)"""
    N = 256
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    F = ra.LocalMatrix(); F.CloneFrom(A)
    F.ILU0Factorize(); F.LUAnalyse()
    rng = np.random.default_rng(3)
    b = ra.LocalVector(data=rng.uniform(-1, 1, n))
    x = ra.LocalVector(); x.Allocate("", n)
    F.LUSolve(b, x)
    # rebuild b = L U x on the host for a sample of 64 complete planes-free rows using the factor entries
    rp, ci, va = F.CopyToCSR()
    xh = x.numpy()
    # u = U x (upper incl. diagonal), then b' = L u (unit lower): do it with numpy on the whole vector
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    upper = ci >= rows
    u = np.bincount(rows[upper], weights=va[upper] * xh[ci[upper]], minlength=n)
    lower = ~upper
    bl = u + np.bincount(rows[lower], weights=va[lower] * u[ci[lower]], minlength=n)
    bh = b.numpy()
    assert np.max(np.abs(bl - bh)) <= 1e-12 * max(1.0, np.max(np.abs(bh))) * 50


def test_mcsgs_forms_agree_at_full_size_256(ra, S):
    N = 256
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    rng = np.random.default_rng(5)
    x = ra.LocalVector(data=rng.uniform(-1, 1, n))
    res = []
    for fused in (True, False):
        pc = S.MultiColoredSGS(); pc.SetFusedSweeps(fused)
        ls = S.BiCGStab(); ls.SetOperator(A); ls.SetPreconditioner(pc); ls.Build()
        assert ls.GetNumColors() == 2
        z = ra.LocalVector(); z.Allocate("", n)
        ls.PrecondApply(x, z)
        res.append(z.numpy().copy())
        ls.Clear()
    assert np.array_equal(res[0], res[1])


def test_iterative_triangular_solve_is_exact_after_level_count_sweeps_256(ra, S):
    N = 256
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    rng = np.random.default_rng(11)
    b = ra.LocalVector(data=rng.uniform(-1, 1, n))
    out = {}
    for key, descr in (("direct", None), ("it_full", (3 * N, 0.0, False)), ("it_tol", (3 * N, 1e-3, True))):
        pc = S.ILU()
        if descr:
            d = S.SolverDescr(); d.SetTriSolverAlg(S.TriSolverAlg_Iterative)
            d.SetIterativeSolverMaxIteration(descr[0]); d.SetIterativeSolverTolerance(descr[1])
            d.EnableIterativeSolverTolerance() if descr[2] else d.DisableIterativeSolverTolerance()
            pc.SetSolverDescriptor(d)
        ls = S.GMRES(); ls.SetOperator(A); ls.SetPreconditioner(pc); ls.Build()
        z = ra.LocalVector(); z.Allocate("", n)
        ra.sync()
        import time
        t = time.perf_counter(); ls.PrecondApply(b, z); ra.sync(); dt = time.perf_counter() - t
        out[key] = (z.numpy().copy(), dt)
        ls.Clear()
    ref = out["direct"][0]
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(out["it_full"][0] - ref)) <= 1e-11 * scale  # nilpotent iteration: exact after 3N-2 sweeps
    # tolerance 1e-3 on the max-norm update/residual figure: close to the exact solve, and stopped long before 3N sweeps
    assert np.max(np.abs(out["it_tol"][0] - ref)) <= 5e-2 * scale
    assert out["it_tol"][1] < 0.25 * out["it_full"][1]


def test_cg_ic_beats_cg_jacobi_256(ra, S):
    N = 256
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
    rhs = ra.LocalVector(); rhs.Allocate("", n); A.Apply(ones, rhs)
    its = {}
    for name, pc in (("jacobi", S.Jacobi()), ("ic", S.IC())):
        ls = S.CG(); ls.SetOperator(A); ls.SetPreconditioner(pc); ls.Init(1e-15, 1e-8, 1e8, 5000); ls.Build()
        x = ra.LocalVector(); x.Allocate("", n)
        ls.Solve(rhs, x)
        assert ls.GetSolverStatus() == 2  # relative tolerance reached
        assert np.linalg.norm(x.numpy() - 1.0) / np.sqrt(n) < 1e-5
        its[name] = ls.GetIterationCount()
        ls.Clear()
    assert its["ic"] < 0.6 * its["jacobi"], its


def test_ilu0_and_ic_build_soak_512(ra, S):
    """the sync-free factorisations and analyses poll other workgroups' results: a rare congestion collapse (20-30 s builds
    or a spin-limit abort at 512^3) was seen once in round 1 and answered with exponential back-off.  Soak: 10 ILU(0) and
    10 IC builds in a row at 512^3, every one inside a wall-time bound, none aborting, and every build giving the
    bit-identical preconditioner (same solve result on the same right-hand side)."""
    import time
    N = 512
    n = N ** 3
    A = ra.LocalMatrix(); A.GenPoisson7(N)
    b = ra.LocalVector(data=np.random.default_rng(1).uniform(-1.0, 1.0, n))
    x = ra.LocalVector(); x.Allocate("", n)
    norms, worst = [], 0.0
    for rep in range(10):
        F = ra.LocalMatrix(); F.CloneFrom(A)
        ra.sync(); t0 = time.perf_counter()
        F.ILU0Factorize(); F.LUAnalyse()
        ra.sync(); dt = time.perf_counter() - t0
        worst = max(worst, dt)
        assert dt < 12.0, (rep, dt)
        F.LUSolve(b, x)
        norms.append(x.Norm())
        del F
    assert len(set(norms)) == 1, norms
    norms = []
    for rep in range(10):
        ls = S.CG(); ls.SetOperator(A); ls.SetPreconditioner(S.IC())
        ra.sync(); t0 = time.perf_counter()
        ls.Build()
        ra.sync(); dt = time.perf_counter() - t0
        worst = max(worst, dt)
        assert dt < 12.0, (rep, dt)
        ls.PrecondApply(b, x)
        norms.append(x.Norm())
        ls.Clear()
    assert len(set(norms)) == 1, norms
    print("worst build %.2f s" % worst)
