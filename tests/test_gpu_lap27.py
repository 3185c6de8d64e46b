"""The reference's own 3-D operator -- the 27-point Laplacian of gen_3d_laplacian (clients/include/utility.hpp:110-177) -- through
the hot path (VERDICT r05 "missing" item 3): the device generator against the host restatement, the SpMV in every format, the
ILU(0) factors, LUSolve, the 8-colour MC-SGS apply -- all bit for bit against the CPU oracle -- and the three solver /
preconditioner pairs of BASELINE.json against the oracle's runs (iteration counts, histories), on 8^3, 16^3 and 33 x 31 x 5
(extents that are no multiples of anything, fewer planes than colours).  Forced variants: the row-pattern product and sweeps
(RAMD_CSR_PAT=1: patterns of up to 27 entries), the general paths (RAMD_CSR_PAT=0)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from rocalution_amd import generators as gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRIDS = [(8, 8, 8), (16, 16, 16), (33, 31, 5)]


@pytest.fixture(scope="module")
def ra():
    import rocalution_amd as ra
    ra.init_rocalution()
    return ra


@pytest.fixture(scope="module")
def S():
    from rocalution_amd import solvers
    return solvers


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), "max abs diff %g" % np.max(np.abs(a - b))


@pytest.mark.parametrize("grid", GRIDS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_generator_spmv_ilu0_lusolve_vs_oracle(ra, oracle, grid, dtype):
    nx, ny, nz = grid
    rp, ci, va = gen.laplace27(nx, ny, nz, dtype)
    n = len(rp) - 1
    G = ra.LocalMatrix(dtype); G.GenLaplace27(nx, ny, nz)
    grp, gci, gva = G.CopyToCSR()
    eq(grp, rp); eq(gci, ci); eq(gva, va)
    x = np.random.default_rng(nx).uniform(-4, 6, n).astype(dtype)
    vx = ra.LocalVector(dtype, data=x)
    ref = oracle.csr_apply(rp, ci, va, x)
    y0 = np.random.default_rng(ny).uniform(-1, 1, n).astype(dtype)
    ref_add = oracle.csr_apply_add(rp, ci, va, x, 0.75, y0.copy())
    for fmt in (ra.CSR, ra.ELL, ra.HYB, ra.COO):
        A = ra.LocalMatrix(dtype); A.GenLaplace27(nx, ny, nz)
        assert A.ConvertTo(fmt) == fmt
        y = ra.LocalVector(dtype); y.Allocate("", n)
        A.Apply(vx, y)
        eq(y.numpy(), ref)
        if fmt in (ra.CSR, ra.ELL, ra.HYB):
            y = ra.LocalVector(dtype, data=y0)
            A.ApplyAdd(vx, 0.75, y)
            eq(y.numpy(), ref_add)
    # ILU(0) factors and LUSolve (host_matrix_csr.cpp:2096-2171, :1163-1221)
    M = ra.LocalMatrix(dtype); M.GenLaplace27(nx, ny, nz)
    M.ILU0Factorize()
    lu = oracle.ilu0(rp, ci, va)
    eq(M.CopyToCSR()[2], lu)
    M.LUAnalyse()
    b = np.random.default_rng(7).uniform(-1, 1, n).astype(dtype)
    y = ra.LocalVector(dtype); y.Allocate("", n)
    for rep in range(2):
        M.LUSolve(ra.LocalVector(dtype, data=b), y)
    eq(y.numpy(), oracle.lusolve(rp, ci, lu, b))


@pytest.mark.parametrize("grid", GRIDS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_eight_colour_mcsgs_apply_vs_oracle(ra, S, oracle, grid, dtype):
    """the multi-colouring of the 27-point operator has 8 colours (a cell and its 26 neighbours differ in at least one parity of
    x, y, z) where the lattice has two cells or more along every axis; the SGS apply in the fused colour sweeps and in the block
    form of the reference (preconditioner_multicolored_gs.cpp:127-215): bit for bit against the oracle"""
    nx, ny, nz = grid
    rp, ci, va = gen.laplace27(nx, ny, nz, dtype)
    n = len(rp) - 1
    rhs = np.random.default_rng(3).uniform(-1, 1, n).astype(dtype)
    want = oracle.precond_apply(oracle.PC_MCSGS, rp, ci, va, rhs)
    A = ra.LocalMatrix(dtype); A.GenLaplace27(nx, ny, nz)
    for fused in (True, False):
        pc = S.MultiColoredSGS(); pc.SetFusedSweeps(fused)
        ls = S.BiCGStab(dtype); ls.SetOperator(A); ls.SetPreconditioner(pc); ls.Build()
        assert ls.GetNumColors() == 8
        z = ra.LocalVector(dtype); z.Allocate("", n)
        for rep in range(2):
            ls.PrecondApply(ra.LocalVector(dtype, data=rhs), z)
        eq(z.numpy(), want)
        ls.Clear()


@pytest.mark.parametrize("grid", GRIDS)
@pytest.mark.parametrize("tag", ["cg_jacobi", "gmres_ilu0", "bicgstab_mcsgs"])
def test_solvers_vs_oracle(ra, S, oracle, grid, tag):
    from test_gpu_solvers import _check_hist, _mk
    nx, ny, nz = grid
    rp, ci, va = gen.laplace27(nx, ny, nz)
    n = len(rp) - 1
    rhs_h = oracle.csr_apply(rp, ci, va, np.ones(n))
    sk = {"cg": oracle.CG, "gmres": oracle.GMRES, "bicgstab": oracle.BICGSTAB}[tag.split("_")[0]]
    pk = {"jacobi": oracle.PC_JACOBI, "ilu0": oracle.PC_ILU0, "mcsgs": oracle.PC_MCSGS}[tag.split("_")[1]]
    ref = oracle.solve(rp, ci, va, rhs_h, solver=sk, precond=pk)
    A = ra.LocalMatrix(); A.GenLaplace27(nx, ny, nz)
    ls = _mk(S, tag); ls.SetOperator(A); ls.Build()
    x = ra.LocalVector(); x.Allocate("", n)
    ls.Solve(ra.LocalVector(data=rhs_h), x)
    assert abs(ls.GetIterationCount() - ref["iters"]) <= 2, (ls.GetIterationCount(), ref["iters"])
    _check_hist(ls.GetResidualHistory(), ref["history"], tag.startswith("bicgstab"), rtol=1e-6)
    assert np.linalg.norm(x.numpy() - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-6


@pytest.mark.parametrize("variant", ["RAMD_CSR_PAT=1", "RAMD_CSR_PAT=0"])
def test_this_file_with_the_row_patterns_forced_on_and_off(variant):
    """row patterns are taken from 2^20 entries on by default; forced on, every matrix of this file runs the pattern product,
    the pattern ELL / HYB products and the pattern colour sweeps with dictionaries of up to 27 entries per row; forced off, the
    paths with the stored columns"""
    env = dict(os.environ)
    env[variant.split("=")[0]] = variant.split("=")[1]
    cmd = [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
           "not forced_on_and_off"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:]


def test_full_size_closed_form_256(ra):
    """256^3 (the bench size): A 1 = 26 - (neighbours - 1) = 27 - count(row) exactly, and A x for integer-valued x against the
    closed form of the stencil (every product and sum exact in fp64: any order gives the same bits); CSR, ELL, HYB"""
    N = 256
    n = N ** 3
    idx = np.arange(n, dtype=np.int64)
    ix, iy, iz = idx % N, (idx // N) % N, idx // (N * N)
    span = lambda i: (i > 0).astype(np.int64) + 1 + (i < N - 1)
    cnt = span(ix) * span(iy) * span(iz)
    xh = ((idx * 7919) % 17 - 8).astype(np.float64)
    X = xh.reshape(N, N, N)
    box = np.zeros_like(X)
    P = np.pad(X, 1)
    for sz in range(3):
        for sy in range(3):
            for sx in range(3):
                box += P[sz:sz + N, sy:sy + N, sx:sx + N]
    want_x = (27.0 * X - box).ravel()  # 26 x_r - sum of the other 26 = 27 x_r - sum of the box
    A = ra.LocalMatrix(); A.GenLaplace27(N)
    assert A.GetNnz() == int(cnt.sum())
    ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
    vx = ra.LocalVector(data=xh)
    y = ra.LocalVector(); y.Allocate("", n)
    for fmt in (ra.CSR, ra.ELL, ra.HYB):
        assert A.ConvertTo(fmt) == fmt
        A.Apply(ones, y)
        assert np.array_equal(y.numpy(), (27 - cnt).astype(np.float64)), fmt
        A.Apply(vx, y)
        assert np.array_equal(y.numpy(), want_x), fmt
