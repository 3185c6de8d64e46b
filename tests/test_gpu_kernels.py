"""GPU parity tests, kernel level: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs, plus the committed golden fixtures (outputs of the genuine rocALUTION host
backend).  Bar: element-wise ops, SpMV in all formats, layouts, ILU(0) factors,
triangular solves, permutations: BIT-EXACT; reductions: relative 1e-13 (tree vs. sequential sum).
"""
import os
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = ["gr3030", "poisson8", "lap2d7", "rand300", "rand300ell", "lap27_6"]  # (lap27_6: the reference's own 27-point operator, 6^3)


@pytest.fixture(scope="module")
def ra():
    import rocalution_amd as ra
    ra.init_rocalution()
    return ra


def _capi():
    from rocalution_amd import capi
    return capi


def _need_offscope(symbol):
    """SPAI / FSAI / RS-AMG / Gershgorin are outside SURVEY.md's scope and behind RAMD_WITH_OFFSCOPE since round 6"""
    if not _capi().has(symbol):
        pytest.skip(symbol + " is not in the default build (RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE)")


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    assert np.array_equal(a, b), "max abs diff %g" % np.max(np.abs(a.astype(np.float64) - b))


def close(a, b, rtol):
    assert abs(a - b) <= rtol * max(abs(a), abs(b), 1e-300), (a, b)


def _mat(ra, g, dtype=np.float64):
    A = ra.LocalMatrix(dtype)
    A.SetDataPtrCSR(g["rowptr"], g["col"], g["val"].astype(dtype))
    return A


@pytest.mark.parametrize("name", CASES)
def test_blas1_vs_golden(ra, name):
    g = load_golden(name)
    x, y, rhs = g["x"], g["y"], g["rhs_ones"]
    vx, vy, vr = ra.LocalVector(data=x), ra.LocalVector(data=y), ra.LocalVector(data=rhs)
    sc = g["blas_scalars"]
    close(vx.Dot(vy), sc[0], 1e-13)
    close(vx.Norm(), sc[2], 1e-13)
    for op, key in ((lambda v: v.AddScale(vy, 0.375), "blas_add_scale"),
                    (lambda v: v.ScaleAdd(-1.25, vy), "blas_scale_add"),
                    (lambda v: v.ScaleAdd2(0.3, vy, -1.7, vr, 0.11), "blas_scale_add2"),
                    (lambda v: v.Scale(1.0 / 3.0), "blas_scale")):
        v = ra.LocalVector(data=x)
        op(v)
        eq(v.numpy(), g[key])
    v = ra.LocalVector(data=np.zeros_like(x))
    v.PointWiseMult(vx, vy)
    eq(v.numpy(), g["blas_pointwise"])


@pytest.mark.parametrize("n", [0, 1, 3, 255, 256, 257, 4099, 1 << 20, (1 << 20) + 3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_blas1_sizes_vs_oracle(ra, oracle, n, dtype):
    rng = np.random.default_rng(n + 7)
    a = rng.uniform(-1, 1, n).astype(dtype)
    b = rng.uniform(-1, 1, n).astype(dtype)
    c = rng.uniform(-1, 1, n).astype(dtype)
    va, vb, vc = (ra.LocalVector(dtype, data=v) for v in (a, b, c))
    if n:
        # exact (fp64) values: the kernels accumulate in fp64 with a fixed-order tree
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        assert abs(va.Dot(vb) - np.dot(a64, b64)) <= 1e-12 * np.abs(a64 * b64).sum() + 1e-300
        close(va.Norm(), float(np.sqrt(np.dot(a64, a64))), 1e-13)
        # the oracle sums like the reference (sequentially, in ValueType): fp32 sums of 1e6 terms are
        # only good to ~1e-3 there, so the oracle comparison is loose for fp32 and tight for fp64
        rtol = 1e-12 if dtype == np.float64 else 2e-3
        assert abs(va.Dot(vb) - float(oracle.dot(a, b))) <= rtol * np.abs(a64 * b64).sum() + 1e-300
        close(va.Norm(), float(oracle.norm(a)), rtol)
        assert abs(va.Asum() - np.abs(a.astype(np.float64)).sum()) <= 1e-6 * max(1, n)
        i, v = va.Amax()
        assert v == np.abs(a).max() and i == int(np.argmax(np.abs(a)))
    else:
        assert va.Dot(vb) == 0.0 and va.Norm() == 0.0
    v = ra.LocalVector(dtype, data=a); v.AddScale(vb, 0.7)
    eq(v.numpy(), oracle.add_scale(a, b, 0.7))
    v = ra.LocalVector(dtype, data=a); v.ScaleAdd(-0.3, vb)
    eq(v.numpy(), oracle.scale_add(a, -0.3, b))
    v = ra.LocalVector(dtype, data=a); v.ScaleAdd2(0.9, vb, 1.1, vc, -2.5)
    eq(v.numpy(), oracle.scale_add2(a, 0.9, b, 1.1, c, -2.5))
    v = ra.LocalVector(dtype, data=a); v.Scale(1.0 / 7.0)
    eq(v.numpy(), oracle.scale(a, 1.0 / 7.0))
    v = ra.LocalVector(dtype, data=a); v.PointWiseMult(vb)
    eq(v.numpy(), a * b)


def test_vector_errors(ra):
    a = ra.LocalVector(data=np.ones(5))
    b = ra.LocalVector(data=np.ones(6))
    with pytest.raises(ra.RamdError):
        a.AddScale(b, 1.0)  # the reference asserts on size mismatch
    with pytest.raises(ra.RamdError):
        a.Dot(b)
    f = ra.LocalVector(np.float32, data=np.ones(5))
    with pytest.raises(ra.RamdError):
        a.ScaleAdd(1.0, f)


def test_cast_and_permute(ra, oracle):
    rng = np.random.default_rng(3)
    n = 1000
    x = rng.uniform(-4, 6, n)
    perm = rng.permutation(n).astype(np.int32)
    vx, vp = ra.LocalVector(data=x), ra.LocalVector(np.int32, data=perm)
    out = ra.LocalVector(); out.Allocate("", n)
    out.CopyFromPermute(vx, vp); eq(out.numpy(), oracle.copy_permute(x, perm))
    out.CopyFromPermuteBackward(vx, vp); eq(out.numpy(), oracle.copy_permute_backward(x, perm))
    f = ra.LocalVector(np.float32); f.CopyFromDouble(vx); eq(f.numpy(), x.astype(np.float32))
    d = ra.LocalVector(np.float64); d.CopyFromFloat(f); eq(d.numpy(), x.astype(np.float32).astype(np.float64))
    idx = rng.integers(0, n, 77).astype(np.int32)
    vi = ra.LocalVector(np.int32, data=idx)
    o = ra.LocalVector(); o.Allocate("", 77)
    vx.GetIndexValues(vi, o); eq(o.numpy(), x[idx])
    o2 = ra.LocalVector(); o2.Allocate("", n); o2.CopyFrom(vx, 10, 20, 100)
    ref = np.zeros(n); ref[20:120] = x[10:110]; eq(o2.numpy(), ref)


@pytest.mark.parametrize("name", CASES)
def test_spmv_all_formats_vs_golden(ra, name):
    g = load_golden(name)
    n = len(g["rowptr"]) - 1
    x, y0 = ra.LocalVector(data=g["x"]), g["y"]
    for fmt, key in ((ra.CSR, "csr"), (ra.ELL, "ell"), (ra.HYB, "hyb"), (ra.COO, "coo")):
        A = _mat(ra, g)
        got = A.ConvertTo(fmt)
        if fmt == ra.ELL and int(g["ell_format"][0]) != ra.ELL:
            assert got == ra.CSR  # refused exactly where the reference refused
            continue
        assert got == fmt
        y = ra.LocalVector(); y.Allocate("", n)
        A.Apply(x, y)
        eq(y.numpy(), g["spmv_" + key])
        y = ra.LocalVector(data=y0)
        A.ApplyAdd(x, -0.75, y)
        eq(y.numpy(), g["spmv_" + key + "_add"])
        if fmt == ra.ELL:
            w, ec, ev = A.ell_arrays()
            assert w == int(g["ell_width"][0])
            eq(ec, g["ell_col"]); eq(ev, g["ell_val"])


@pytest.mark.parametrize("name", CASES)
def test_csr_matrix_algebra_vs_golden(ra, name):
    """Transpose, MatrixMult (A * A^T), MatrixAdd with subset and union patterns: arrays bit-exact vs the genuine library"""
    g = load_golden(name)
    A = _mat(ra, g)

    def check(M, tag):
        rp, ci, va = M.CopyToCSR()
        eq(rp, g[tag + "_rowptr"]); eq(ci, g[tag + "_col"]); eq(va, g[tag + "_val"])

    T = ra.LocalMatrix(); T.CloneFrom(A); T.Transpose()
    check(T, "alg_transpose")
    T2 = ra.LocalMatrix(); A.Transpose(T2)
    check(T2, "alg_transpose")
    AA = ra.LocalMatrix(); AA.MatrixMult(A, T)
    check(AA, "alg_matmult")
    S_ = ra.LocalMatrix(); S_.CloneFrom(AA); S_.MatrixAdd(A, 0.5, -2.0, False)
    check(S_, "alg_add_subset")
    U = ra.LocalMatrix(); U.CloneFrom(A); U.MatrixAdd(AA, 1.5, 0.25, True)
    check(U, "alg_add_union")
    # Sort: rows scrambled on the host come back sorted, values following their columns
    rp, ci, va = g["rowptr"], g["col"].copy(), g["val"].copy()
    rng = np.random.default_rng(5)
    for i in range(len(rp) - 1):
        p = rng.permutation(rp[i + 1] - rp[i]) + rp[i]
        ci[rp[i]:rp[i + 1]] = ci[p]; va[rp[i]:rp[i + 1]] = va[p]
    B = ra.LocalMatrix(); B.SetDataPtrCSR(rp, ci, va); B.Sort()
    brp, bci, bva = B.CopyToCSR()
    eq(bci, g["col"]); eq(bva, g["val"])
    # y = (A A^T) x through the product equals A (A^T x) to round-off
    n = len(rp) - 1
    x = ra.LocalVector(data=g["x"])
    y1 = ra.LocalVector(); y1.Allocate("", n); AA.Apply(x, y1)
    t = ra.LocalVector(); t.Allocate("", n); T.Apply(x, t)
    y2 = ra.LocalVector(); y2.Allocate("", n); A.Apply(t, y2)
    assert np.allclose(y1.numpy(), y2.numpy(), rtol=1e-12, atol=1e-10 * np.max(np.abs(y2.numpy())))


@pytest.mark.parametrize("name", CASES)
def test_spai_matrix_vs_golden(ra, name):
    """SPAI: per-row least squares by Householder QR on the device -- pattern identical, values identical on the
    integer-valued operators and within 1e-13 on the random ones"""
    _need_offscope("ramd_mat_spai")
    g = load_golden(name)
    A = _mat(ra, g)
    A.SPAI()
    rp, ci, va = A.CopyToCSR()
    eq(rp, g["spai_M_rowptr"]); eq(ci, g["spai_M_col"])
    if name.startswith("rand"):
        # general real values: the installed 4.1 library is one ulp away from the loops of the 3.2 source in 30 % of the
        # entries (a host restatement of that source shows the same one-ulp picture), so: 1e-13 relative
        assert np.allclose(va, g["spai_M_val"], rtol=1e-13, atol=1e-15)
    else:
        eq(va, g["spai_M_val"])


@pytest.mark.parametrize("name", CASES)
def test_ilup_factors_vs_golden(ra, name):
    """ILU(p): fill levels on the pattern of A^(p+1) (p = 1, 2) and ILU(0) on that whole pattern (level=False) --
    pattern and values identical to the genuine library's host factorisation"""
    g = load_golden(name)
    for key, p, level in (("ilu1", 1, True), ("ilu2", 2, True), ("ilu1n", 1, False)):
        LU = _mat(ra, g)
        LU.ILUpFactorize(p, level)
        rp, ci, va = LU.CopyToCSR()
        eq(rp, g[key + "_rowptr"]); eq(ci, g[key + "_col"]); eq(va, g[key + "_val"])


@pytest.mark.parametrize("p", [1, 2])
def test_ilup_product_property_poisson48(ra, p):
    """size-independent property of an incomplete factorisation with fill levels: (L U)_ij = a_ij on the retained
    pattern (0 on the fill entries), checked on the 48^3 Poisson operator with the host's sparse product"""
    import scipy.sparse as sp
    A = ra.LocalMatrix(); A.GenPoisson7(48)
    n = A.GetM()
    arp, aci, ava = A.CopyToCSR()
    F = ra.LocalMatrix(); F.CloneFrom(A)
    F.ILUpFactorize(p)
    rp, ci, va = F.CopyToCSR()
    assert len(ci) > len(aci) and np.all(np.diff(rp) > 0)
    Fm = sp.csr_matrix((va, ci, rp), shape=(n, n))
    L = sp.tril(Fm, -1).tocsr() + sp.identity(n, format="csr")
    U = sp.triu(Fm, 0).tocsr()
    P = (L @ U).tocsr()
    Am = sp.csr_matrix((ava, aci, arp), shape=(n, n))
    mask = sp.csr_matrix((np.ones(len(ci)), ci, rp), shape=(n, n))
    D = (P - Am).multiply(mask)
    assert abs(D).max() < 1e-12
    # the factor has more entries than the pattern of A and fewer than the whole power pattern
    S = Am.copy(); S.data[:] = 1.0
    Sp = S
    for _ in range(p):
        Sp = (Sp @ S).tocsr()
    assert len(ci) <= Sp.nnz


def test_ilup_thread_per_row_path_in_a_fresh_process():
    """rows of the power pattern beyond 256 entries take the thread-per-row sweep; forced here (switch read once per
    process) on the golden matrices -- same arrays"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x",
                        "-m", "gpu", "-k", "ilup_factors_vs_golden"], env=dict(os.environ, RAMD_ILUP_WAVE="0"),
                       cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and ("%d passed" % len(CASES)) in out, out[-3000:]


def test_fsai_factor_vs_golden(ra):
    """FSAI(1): per-row dense LU on the lower pattern and the scaling -- factor arrays identical to the genuine library"""
    _need_offscope("ramd_mat_fsai")
    for name in ("gr3030", "poisson8", "lap2d7"):
        g = load_golden(name)
        A = _mat(ra, g)
        A.FSAI(1)
        rp, ci, va = A.CopyToCSR()
        eq(rp, g["fsai_G_rowptr"]); eq(ci, g["fsai_G_col"]); eq(va, g["fsai_G_val"])
        A2 = _mat(ra, g)
        A2.FSAI(2)  # lower part of the pattern of A^2
        rp, ci, va = A2.CopyToCSR()
        eq(rp, g["fsai2_G_rowptr"]); eq(ci, g["fsai2_G_col"]); eq(va, g["fsai2_G_val"])
        pat = _mat(ra, g); pat.ILUpFactorize(1)  # an external pattern: the ILU(1) factor's
        A3 = _mat(ra, g)
        A3.FSAI(1, pat)
        rp, ci, va = A3.CopyToCSR()
        eq(rp, g["fsai3_G_rowptr"]); eq(ci, g["fsai3_G_col"]); eq(va, g["fsai3_G_val"])


@pytest.mark.parametrize("lds", ["1", "0", "0+chunks"])
def test_matmult_long_row_paths_in_a_fresh_process(lds):
    """MatrixMult leaves the per-thread insertion when a row has many products: (lds=1) one workgroup per row sorts the
    products by (column, generation index) in LDS, (lds=0 / rows beyond 2048 products) two global stable sorts; both
    followed by in-order run sums.  Forced here on the small golden matrices (the switches are read once per
    process) -- same bit-exact arrays as the reference"""
    import os
    import subprocess
    import sys
    env = dict(os.environ, RAMD_MM_INSERT_LIMIT="0", RAMD_MM_LDS=lds[0])
    if lds.endswith("chunks"):  # the global sort runs over row chunks of bounded product count: force many chunks
        env["RAMD_MM_CHUNK"] = "700"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_kernels.py"), "-q", "-x",
                        "-m", "gpu", "-k", "csr_matrix_algebra_vs_golden"], env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0 and ("%d passed" % len(CASES)) in out, out[-3000:]


@pytest.mark.parametrize("name", ["gr3030", "poisson8", "lap2d7"])
def test_amg_pmis_aggregation_vs_golden(ra, name):
    """UA-AMG setup of the first level: strong connections, PMIS aggregates, root nodes, prolongation operator --
    every array identical to the genuine library's (host backend)"""
    g = load_golden(name)
    A = _mat(ra, g)
    conn, agg, roots = A.AMGPMISAggregate(0.01)
    eq(conn.numpy(), g["amg_conn"]); eq(agg.numpy(), g["amg_agg"]); eq(roots.numpy(), g["amg_roots"])
    P = ra.LocalMatrix()
    A.AMGUnsmoothedAggregation(agg, roots, P)
    rp, ci, va = P.CopyToCSR()
    eq(rp, g["amg_P_rowptr"]); eq(ci, g["amg_P_col"]); eq(va, g["amg_P_val"])
    assert P.GetN() == int(g["amg_P_col"].max()) + 1
    # the default strategy: the reference's sequential greedy sweep, restated as a sync-free device sweep
    gconn, gagg, groots = A.AMGGreedyAggregate(0.01)
    eq(gconn.numpy(), g["amg_conn"]); eq(gagg.numpy(), g["amg_gagg"]); eq(groots.numpy(), g["amg_groots"])
    # classical AMG: PMIS C/F splitting (hash + strong in-degree weights) and direct interpolation (out of scope: only in a
    # library built with -DRAMD_WITH_OFFSCOPE)
    if _capi().has("ramd_mat_rs_pmis_coarsening"):
        cf, S_ = A.RSPMISCoarsening(0.25)
        eq(cf.numpy(), g["rs_cf"]); eq(S_.numpy(), g["rs_S"])
        Prs = ra.LocalMatrix()
        A.RSDirectInterpolation(cf, S_, Prs)
        rp, ci, va = Prs.CopyToCSR()
        eq(rp, g["rs_P_rowptr"]); eq(ci, g["rs_P_col"]); eq(va, g["rs_P_val"])
    # smoothed aggregation: (I - relax D_f^-1 A_f) P_tent, both lumping strategies
    for key, relax, lump in (("amg_Ps", 2.0 / 3.0, 0), ("amg_Ps1", 0.5, 1)):
        Ps = ra.LocalMatrix()
        A.AMGSmoothedAggregation(relax, conn, agg, roots, Ps, lump)
        rp, ci, va = Ps.CopyToCSR()
        eq(rp, g[key + "_rowptr"]); eq(ci, g[key + "_col"]); eq(va, g[key + "_val"])


@pytest.mark.parametrize("N", [5, 16, 33])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_spmv_poisson_vs_oracle(ra, oracle, N, dtype):
    from rocalution_amd import generators as gen
    rp, ci, va = gen.poisson7(N, dtype)
    n = len(rp) - 1
    x = np.random.default_rng(N).uniform(-4, 6, n).astype(dtype)
    vx = ra.LocalVector(dtype, data=x)
    ref = oracle.csr_apply(rp, ci, va, x)
    for fmt in (ra.CSR, ra.ELL, ra.HYB, ra.COO):
        A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va)
        assert A.ConvertTo(fmt) == fmt
        y = ra.LocalVector(dtype); y.Allocate("", n)
        A.Apply(vx, y)
        eq(y.numpy(), ref)
    # device generator == host generator
    G = ra.LocalMatrix(dtype); G.GenPoisson7(N)
    grp, gci, gva = G.CopyToCSR()
    eq(grp, rp); eq(gci, ci); eq(gva, va)


def test_spmv_edge_cases(ra, oracle):
    # empty matrix: Apply zero-fills, ApplyAdd is a no-op (local_matrix.cpp:2176-2209)
    A = ra.LocalMatrix()
    A.SetDataPtrCSR(np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0))
    x = ra.LocalVector(data=np.ones(5)); y = ra.LocalVector(data=np.full(5, 3.0))
    A.ApplyAdd(x, 2.0, y); eq(y.numpy(), np.full(5, 3.0))
    A.Apply(x, y); eq(y.numpy(), np.zeros(5))
    # ragged rows incl. empty rows and one very long row (several LDS chunks)
    rng = np.random.default_rng(5)
    n = 700
    rows = [sorted(set(rng.integers(0, n, rng.integers(0, 9)).tolist())) for _ in range(n)]
    rows[13] = list(range(n))
    rows[400] = []
    rows[401] = list(range(0, n, 2))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    ci = np.array([c for r in rows for c in r], np.int32)
    va = rng.uniform(-1, 1, len(ci))
    xv = rng.uniform(-1, 1, n)
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    y = ra.LocalVector(); y.Allocate("", n)
    A.Apply(ra.LocalVector(data=xv), y)
    eq(y.numpy(), oracle.csr_apply(rp, ci, va, xv))
    with pytest.raises(ra.RamdError):
        A.Apply(ra.LocalVector(data=np.ones(n + 1)), y)  # size mismatch: the reference asserts


@pytest.mark.parametrize("name", CASES)
def test_diag_ilu_trisolve_vs_golden(ra, name):
    g = load_golden(name)
    n = len(g["rowptr"]) - 1
    A = _mat(ra, g)
    d = ra.LocalVector(); A.ExtractInverseDiagonal(d)
    eq(d.numpy(), g["inv_diag"])
    LU = ra.LocalMatrix(); LU.CloneFrom(A)
    LU.ILU0Factorize()
    rp, ci, va = LU.CopyToCSR()
    eq(rp, g["ilu0_rowptr"]); eq(ci, g["ilu0_col"]); eq(va, g["ilu0_val"])
    LU.LUAnalyse()
    x = ra.LocalVector(data=g["x"]); y = ra.LocalVector(); y.Allocate("", n)
    LU.LUSolve(x, y)
    eq(y.numpy(), g["lusolve"])
    LU.LUSolve(x, y)  # plans are reusable
    eq(y.numpy(), g["lusolve"])
    A.LAnalyse(False); A.LSolve(x, y); eq(y.numpy(), g["lsolve_nonunit"])
    A.UAnalyse(False); A.USolve(x, y); eq(y.numpy(), g["usolve_nonunit"])


@pytest.mark.parametrize("name", CASES)
def test_multicoloring_permute_vs_golden(ra, name):
    g = load_golden(name)
    A = _mat(ra, g)
    nc, sizes, perm = A.MultiColoring()
    assert nc == int(g["mc_num_colors"][0])
    eq(sizes, g["mc_sizes"]); eq(perm.numpy(), g["mc_perm"])
    A.Permute(perm)
    rp, ci, va = A.CopyToCSR()
    eq(rp, g["permuted_rowptr"]); eq(ci, g["permuted_col"]); eq(va, g["permuted_val"])


def test_extract_submatrix_vs_oracle(ra, oracle):
    g = load_golden("gr3030")
    A = _mat(ra, g)
    for (r0, c0, rs, cs) in ((0, 0, 900, 900), (100, 300, 250, 411), (899, 0, 1, 900), (5, 5, 0, 0)):
        S = ra.LocalMatrix()
        A.ExtractSubMatrix(r0, c0, rs, cs, S)
        rp, ci, va = S.CopyToCSR()
        orp, oci, ova = oracle.extract_submatrix(g["rowptr"], g["col"], g["val"], r0, c0, rs, cs)
        eq(rp, orp); eq(ci, oci); eq(va, ova)


@pytest.mark.parametrize("N", [24, 40, 77, 100, 168])
def test_ilu_lusolve_poisson_vs_oracle(ra, oracle, N):
    """deep dependency DAG (3N-2 levels), exercised repeatedly to shake out stale hand-offs.  N = 77, 100: grid lines that are
    no multiple of the tile box (boundary tiles, upper and lower tiles that do not coincide: the 4-byte index lists fall back to
    pairs where a tile's sources span too much).  N = 168: 4.7 M rows = 9261
    box tiles on ~2000 persistent workgroups, so every workgroup walks several tiles (LDS ring reuse, per-tile write-back
    of the natural-order output, all 16 ticket streams) -- still bit for bit"""
    from rocalution_amd import generators as gen
    rp, ci, va = gen.poisson7(N)
    n = len(rp) - 1
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    A.ILU0Factorize()
    lu = oracle.ilu0(rp, ci, va)
    eq(A.CopyToCSR()[2], lu)
    A.LUAnalyse()
    rng = np.random.default_rng(N)
    y = ra.LocalVector(); y.Allocate("", n)
    for rep in range(5):
        b = rng.uniform(-1, 1, n)
        A.LUSolve(ra.LocalVector(data=b), y)
        eq(y.numpy(), oracle.lusolve(rp, ci, lu, b))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("reach,side", [(3, 3), (40, 3), (700, 3), (40, 7)])
def test_ilu0_rows_in_registers_irregular_vs_oracle(ra, oracle, dtype, reach, side):
    """ILU(0) of rows with at most 8 entries runs in natural order with the row in registers; pivot rows held by a lower
    lane of the same wave come from that lane's registers (k_ilu0_rows).  Random unsymmetric patterns whose lower entries
    reach 3 / 40 / 700 rows back: in-wave pivots at every lane distance, several per row, mixed with out-of-wave ones, a
    missing diagonal neighbour here and there -- factors bit for bit, fp64 and fp32"""
    rng = np.random.default_rng(reach)
    n = 1 << 20  # 4096 blocks of 256 rows: the hyperplane block order is in use
    rows, cols = [], []
    i = np.arange(n)
    for s in range(side):  # (side = 7: rows of up to 15 entries, the 16-entry form of the kernel)
        d = rng.integers(1, reach + 1, n)
        keep = rng.random(n) < 0.8
        lo = i - d
        ok = keep & (lo >= 0)
        rows.append(i[ok]); cols.append(lo[ok])
        d = rng.integers(1, reach + 1, n)
        keep = rng.random(n) < 0.8
        hi = i + d
        ok = keep & (hi < n)
        rows.append(i[ok]); cols.append(hi[ok])
    rows.append(i); cols.append(i)
    import scipy.sparse as sp
    r = np.concatenate(rows); c = np.concatenate(cols)
    P = sp.csr_matrix((np.ones(len(r)), (r, c)), shape=(n, n))
    P.sum_duplicates(); P.sort_indices()
    assert np.diff(P.indptr).max() <= 2 * side + 1
    P.data[:] = rng.uniform(-1.0, 1.0, len(P.data))
    diag = np.asarray(abs(P).sum(axis=1)).ravel() + 1.0
    A = (P - sp.diags(P.diagonal()) + sp.diags(diag)).tocsr(); A.sort_indices()
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)
    M = ra.LocalMatrix(dtype=dtype)
    M.SetDataPtrCSR(rp, ci, va)
    M.ILU0Factorize()
    eq(M.CopyToCSR()[2], oracle.ilu0(rp, ci, va))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("width", [8, 16])
def test_ilu0_rows_without_an_entry_at_or_right_of_the_diagonal(ra, oracle, dtype, width):
    """a row of EXACTLY 8 (16) entries, all of them left of the diagonal, has its pivot position one past its end: it needs
    one register slot more than it has entries (the 8-entry form would drop its last column and write a foreign value
    slot).  Such rows take the next wider form / the level-order sweep; factors bit for bit.  (No later row uses such a
    row as a pivot row: the reference reads past the end of the row there.)"""
    import scipy.sparse as sp
    rng = np.random.default_rng(width)
    n = 1 << 14
    A = sp.diags([rng.uniform(-1, 0, n - 2), rng.uniform(-1, 0, n - 1), np.full(n, 5.0), rng.uniform(-1, 0, n - 1)],
                 [-2, -1, 0, 1]).tolil()
    odd = list(range(100, n - 100, 517))
    for r in odd:
        A[:, r] = 0.0  # nobody depends on row r
        A[r, :] = 0.0
        A[r, r - width:r] = rng.uniform(-1, -0.1, width)  # `width` entries, none at or right of the diagonal
    A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)
    assert np.diff(rp).max() == width and all(ci[rp[r + 1] - 1] < r for r in odd)
    M = ra.LocalMatrix(dtype=dtype)
    M.SetDataPtrCSR(rp, ci, va)
    M.ILU0Factorize()
    eq(M.CopyToCSR()[2], oracle.ilu0(rp, ci, va))


def test_ilu_lusolve_27_point_stencil_vs_oracle(ra, oracle):
    """13 strictly-lower entries per row: the eight-lanes-per-row form of the box-tile solve on cubic tiles (the shell
    surrogate exercises it on 2-D parallelograms), ILU(0) by the wave-per-row sweep -- factors and solutions bit for bit"""
    import scipy.sparse as sp
    N = 36
    t = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
    P = sp.kron(sp.kron(t, t), t).tocsr()  # 27-point pattern
    P.data[:] = -1.0
    A = (P + sp.diags(np.full(N ** 3, 28.0))).tocsr()  # diagonal 27 (= 28 - 1): strictly dominant
    A.sort_indices()
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    n = N ** 3
    M = ra.LocalMatrix(); M.SetDataPtrCSR(rp, ci, va)
    M.ILU0Factorize()
    lu = oracle.ilu0(rp, ci, va)
    eq(M.CopyToCSR()[2], lu)
    M.LUAnalyse()
    rng = np.random.default_rng(27)
    y = ra.LocalVector(); y.Allocate("", n)
    for rep in range(3):
        b = rng.uniform(-1, 1, n)
        M.LUSolve(ra.LocalVector(data=b), y)
        eq(y.numpy(), oracle.lusolve(rp, ci, lu, b))


@pytest.mark.parametrize("name", CASES)
def test_convert_back_to_csr_and_clone(ra, name):
    """ELL/HYB/COO -> CSR (host_conversion.cpp:690-960) restores the CSR arrays (golden matrices have
    sorted rows), X -> Y goes through CSR (local_matrix.cpp:2085-2093), CloneFrom works in any format"""
    g = load_golden(name)
    n = len(g["rowptr"]) - 1
    x = ra.LocalVector(data=g["x"])
    for fmt in (ra.ELL, ra.HYB, ra.COO):
        A = _mat(ra, g)
        if A.ConvertTo(fmt) != fmt:
            continue
        B = ra.LocalMatrix(); B.CloneFrom(A)
        assert B.GetFormat() == fmt
        y = ra.LocalVector(); y.Allocate("", n)
        B.Apply(x, y)
        eq(y.numpy(), g["spmv_" + {ra.ELL: "ell", ra.HYB: "hyb", ra.COO: "coo"}[fmt]])
        assert B.ConvertTo(ra.CSR) == ra.CSR
        rp, ci, va = B.CopyToCSR()
        eq(rp, g["rowptr"]); eq(ci, g["col"]); eq(va, g["val"])
        other = ra.HYB if fmt != ra.HYB else ra.COO
        assert A.ConvertTo(other) == other  # X -> CSR -> Y
        A.Apply(x, y)
        eq(y.numpy(), g["spmv_hyb"] if other == ra.HYB else g["spmv_coo"])


def test_coo_to_csr_sorts_columns(ra):
    # rows with unsorted columns: COO keeps storage order, coo_to_csr sorts inside each row (stable)
    rng = np.random.default_rng(11)
    n = 300
    rows = [rng.permutation(n)[:rng.integers(0, 12)].tolist() for _ in range(n)]
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    ci = np.array([c for r in rows for c in r], np.int32)
    va = rng.uniform(-1, 1, len(ci))
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    assert A.ConvertTo(ra.COO) == ra.COO
    assert A.ConvertTo(ra.CSR) == ra.CSR
    rp2, ci2, va2 = A.CopyToCSR()
    eq(rp2, rp)
    for i in range(n):
        o = np.argsort(ci[rp[i]:rp[i + 1]], kind="stable")
        eq(ci2[rp[i]:rp[i + 1]], ci[rp[i]:rp[i + 1]][o])
        eq(va2[rp[i]:rp[i + 1]], va[rp[i]:rp[i + 1]][o])
    # HYB -> CSR keeps the row order (ELL slots, then the COO tail) even when unsorted
    B = ra.LocalMatrix(); B.SetDataPtrCSR(rp, ci, va)
    assert B.ConvertTo(ra.HYB) == ra.HYB
    assert B.ConvertTo(ra.CSR) == ra.CSR
    rp3, ci3, va3 = B.CopyToCSR()
    eq(rp3, rp); eq(ci3, ci); eq(va3, va)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fused_apply_add_dot(ra, oracle, dtype):
    """ghost-part ApplyAdd with the dot correction: y bit-exact with ApplyAdd, slot == <p, y_new>"""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(3)
    n, nc = 5000, 700
    touched = np.sort(rng.choice(n, 900, replace=False))
    cnt = np.zeros(n, np.int64); cnt[touched] = rng.integers(1, 4, len(touched))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum(cnt)
    ci = rng.integers(0, nc, rp[-1]).astype(np.int32)
    va = rng.uniform(-1, 1, rp[-1]).astype(dtype)
    xr = rng.uniform(-1, 1, nc).astype(dtype)
    y0 = rng.uniform(-1, 1, n).astype(dtype)
    p = rng.uniform(-1, 1, n).astype(dtype)
    for fmt in (ra.COO, ra.CSR):  # COO: fused kernel; CSR: ApplyAdd + full dot
        G = ra.LocalMatrix(dtype); G.SetDataPtrCSR(rp, ci, va, nrow=n, ncol=nc)
        assert G.ConvertTo(fmt) == fmt
        y = ra.LocalVector(dtype, data=y0); yref = ra.LocalVector(dtype, data=y0)
        vp, vx = ra.LocalVector(dtype, data=p), ra.LocalVector(dtype, data=xr)
        slot = 5
        capi.check(lib.ramd_scalars_set(slot, float(np.dot(p.astype(np.float64), y0.astype(np.float64)))))
        capi.check(lib.ramd_fused_apply_add_dot(G._h, vx._h, 1.0, y._h, vp._h, slot))
        G.ApplyAdd(vx, 1.0, yref)
        eq(y.numpy(), yref.numpy())
        out = (C.c_double * 1)()
        capi.check(lib.ramd_scalars_fetch(out, slot, 1))
        exact = float(np.dot(p.astype(np.float64), y.numpy().astype(np.float64)))
        close(out[0], exact, 1e-12 if dtype == np.float64 else 1e-6)


def _sym_pattern(n, deg, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    r = rng.integers(0, n, n * deg); c = rng.integers(0, n, n * deg)
    M = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(n, n))
    M = ((M + M.T + sp.eye(n)) > 0).astype(np.float64).tocsr()
    M.sort_indices()
    return M.indptr.astype(np.int32), M.indices.astype(np.int32), np.arange(1, M.nnz + 1, dtype=np.float64)


@pytest.mark.parametrize("case", ["poisson24", "sym5000", "sym_dense_rows", "lap2d", "overflow80", "unsym"])
def test_multicoloring_device_vs_oracle(ra, oracle, case):
    """the device sweep (symmetric patterns), the > 64 colour fallback and the unsymmetric fallback all give
    the colours of the serial reference sweep"""
    from rocalution_amd import generators as gen
    if case == "poisson24":
        rp, ci, va = gen.poisson7(24)
    elif case == "sym5000":
        rp, ci, va = _sym_pattern(5000, 3, 1)
    elif case == "sym_dense_rows":
        rp, ci, va = _sym_pattern(600, 12, 2)
    elif case == "lap2d":
        rp, ci, va = gen.laplace2d(70)
    elif case == "overflow80":
        n = 80
        rp = (np.arange(n + 1) * n).astype(np.int32); ci = np.tile(np.arange(n, dtype=np.int32), n)
        va = np.ones(n * n)
    else:
        rp, ci, va = gen.random_sparse(700, 5, seed=9)
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    nc, sizes, perm = A.MultiColoring()
    onc, osizes, operm = oracle.multicoloring(rp, ci)
    assert nc == onc
    eq(sizes, osizes); eq(perm.numpy(), operm)


def test_file_io_device_objects(ra, tmp_path):
    """file IO on accelerator-resident objects (any format) against the files of the genuine library"""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden", "io")
    exp = np.load(os.path.join(gold, "io_expected.npz"))
    A = ra.LocalMatrix(); A.ReadFileCSR(os.path.join(gold, "ref_A.csr"))
    rp, ci, va = A.CopyToCSR()
    eq(rp, exp["rowptr"]); eq(ci, exp["col"]); eq(va, exp["val"])
    ref_mtx = open(os.path.join(gold, "ref_A.mtx"), "rb").read()
    for fmt in (ra.CSR, ra.ELL, ra.HYB, ra.COO):
        B = ra.LocalMatrix(); B.CloneFrom(A)
        assert B.ConvertTo(fmt) == fmt
        f = str(tmp_path / ("A_%d.mtx" % fmt))
        B.WriteFileMTX(f)
        assert open(f, "rb").read() == ref_mtx
        assert B.GetFormat() == fmt  # writing does not change the object
    x = ra.LocalVector(); x.ReadFileBinary(os.path.join(gold, "ref_x.bin"))
    eq(x.numpy(), exp["x"])
    f = str(tmp_path / "x.dat"); x.WriteFileASCII(f)
    assert open(f, "rb").read() == open(os.path.join(gold, "ref_x.dat"), "rb").read()
    y = ra.LocalVector(); y.ReadFileASCII(f)
    assert np.allclose(y.numpy(), exp["x"], rtol=1e-6)


@pytest.mark.parametrize("name", ["gr3030", "rand300", "rand300ell"])
def test_fused_apply_dot_all_formats(ra, name):
    """y = A x with <x,y> in the same pass: y bit-exact with Apply in every format (HYB incl. its COO tail)"""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    g = load_golden(name)
    n = len(g["rowptr"]) - 1
    x = ra.LocalVector(data=g["x"])
    for fmt, key in ((ra.CSR, "csr"), (ra.ELL, "ell"), (ra.HYB, "hyb"), (ra.COO, "coo")):
        A = _mat(ra, g)
        if A.ConvertTo(fmt) != fmt:
            continue
        y = ra.LocalVector(); y.Allocate("", n)
        capi.check(lib.ramd_fused_apply_dot(A._h, x._h, y._h, 7))
        eq(y.numpy(), g["spmv_" + key])
        out = (C.c_double * 1)()
        capi.check(lib.ramd_scalars_fetch(out, 7, 1))
        close(out[0], float(np.dot(g["x"], y.numpy())), 1e-12)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("precond", [False, True])
def test_fused_bicgstab_updates(ra, dtype, precond):
    """the three BiCGStab update kernels against the unfused vector ops (bit-exact vectors), incl. the
    omega breakdown branch"""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(17)
    n = 100003
    mk = lambda: rng.uniform(-1, 1, n).astype(dtype)
    x, d, sv, r, t, r0, p, q = (mk() for _ in range(8))
    rho, r0q, tr, tt = 0.731, -1.917, 0.377, 2.113
    S_TR, S_R0Q, S_RHO, S_RR, S_NEW, S_FLAG = 0, 2, 3, 4, 5, 6
    T = dtype

    def setslots(tr_):
        for s_, v_ in ((S_RHO, rho), (S_R0Q, r0q), (S_TR, tr_), (S_TR + 1, tt)):
            capi.check(lib.ramd_scalars_set(s_, float(v_)))

    V = lambda a: ra.LocalVector(dtype, data=a)
    alpha = T(T(rho) / T(r0q)); omega = T(T(tr) / T(tt))
    # r_update == AddScale(q, -alpha)
    setslots(tr)
    vr, vq = V(r), V(q)
    capi.check(lib.ramd_fused_bicg_r_update(vr._h, vq._h, S_RHO, S_R0Q))
    ref = V(r); ref.AddScale(vq, float(-alpha))
    eq(vr.numpy(), ref.numpy())
    # xr_update == ScaleAdd2 + AddScale + two dots
    vx, vd, vs, vr, vt, v0, vp = V(x), V(d), V(sv), V(r), V(t), V(r0), V(p)
    capi.check(lib.ramd_fused_bicg_xr_update(vx._h, vd._h if precond else None, vs._h if precond else None, vr._h,
                                             vt._h, v0._h, vp._h, S_RHO, S_R0Q, S_TR, S_RR, S_NEW, S_FLAG))
    rx, rr_ = V(x), V(r)
    if precond:
        rx.ScaleAdd2(1.0, vd, float(alpha), vs, float(omega))
    else:
        rx.ScaleAdd2(1.0, vp, float(alpha), V(r), float(omega))
    rr_.AddScale(vt, float(-omega))
    eq(vx.numpy(), rx.numpy()); eq(vr.numpy(), rr_.numpy())
    out = (C.c_double * 7)()
    capi.check(lib.ramd_scalars_fetch(out, 0, 7))
    rn = rr_.numpy().astype(np.float64)
    tol = 1e-12 if dtype == np.float64 else 1e-6
    close(out[S_RR], float(rn @ rn), tol); close(out[S_NEW], float(r0.astype(np.float64) @ rn), tol)
    assert out[S_FLAG] == 0.0
    # direction == ScaleAdd2(beta, q, -beta*omega, r, 1) with beta from the record
    beta = T(T(T(out[S_NEW]) / T(rho)) * T(alpha / omega))
    vp2 = V(p)
    capi.check(lib.ramd_fused_bicg_direction(vp2._h, vq._h, vr._h, S_RHO, S_R0Q, S_TR, S_NEW))
    rp_ = V(p); rp_.ScaleAdd2(float(beta), vq, float(T(-beta) * omega), vr, 1.0)
    eq(vp2.numpy(), rp_.numpy())
    # breakdown: <t,r> = 0 -> omega = 0 -> only x += alpha p, flag raised, r untouched
    setslots(0.0)
    vx, vr = V(x), V(r)
    capi.check(lib.ramd_fused_bicg_xr_update(vx._h, vd._h if precond else None, vs._h if precond else None, vr._h,
                                             vt._h, v0._h, vp._h, S_RHO, S_R0Q, S_TR, S_RR, S_NEW, S_FLAG))
    rx = V(x); rx.AddScale(vp, float(alpha))
    eq(vx.numpy(), rx.numpy()); eq(vr.numpy(), r)
    capi.check(lib.ramd_scalars_fetch(out, 0, 7))
    assert out[S_FLAG] == 1.0


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csr_long_rows_across_lds_chunks(ra, oracle, dtype):
    """rows of 20-50 entries: a 256-row workgroup needs several LDS chunks, rows straddle chunk borders;
    empty rows and one row with every column included.  Apply, ApplyAdd, fused dot: bit-exact."""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(31)
    n = 3000
    rows = []
    for i in range(n):
        k = int(rng.integers(20, 50))
        rows.append(sorted(set(np.clip(i + rng.integers(-400, 401, k), 0, n - 1).tolist())))
    rows[7] = []; rows[8] = []; rows[1500] = list(range(0, n, 1))
    rp = np.zeros(n + 1, np.int32); rp[1:] = np.cumsum([len(r) for r in rows])
    ci = np.array([c for r in rows for c in r], np.int32)
    va = rng.uniform(-1, 1, len(ci)).astype(dtype)
    xh = rng.uniform(-1, 1, n).astype(dtype); y0 = rng.uniform(-1, 1, n).astype(dtype)
    A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va)
    x = ra.LocalVector(dtype, data=xh)
    y = ra.LocalVector(dtype); y.Allocate("", n)
    A.Apply(x, y)
    ref = oracle.csr_apply(rp, ci, va, xh)
    eq(y.numpy(), ref)
    ya = ra.LocalVector(dtype, data=y0)
    A.ApplyAdd(x, -0.75, ya)
    eq(ya.numpy(), oracle.csr_apply_add(rp, ci, va, xh, dtype(-0.75), y0))
    w = ra.LocalVector(dtype); w.Allocate("", n)
    capi.check(lib.ramd_fused_apply_dot(A._h, x._h, w._h, 11))
    eq(w.numpy(), ref)
    out = (C.c_double * 1)()
    capi.check(lib.ramd_scalars_fetch(out, 11, 1))
    close(out[0], float(np.dot(xh.astype(np.float64), ref.astype(np.float64))), 1e-12 if dtype == np.float64 else 1e-6)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_matrix_utilities_vs_host_loops(ra, dtype):
    """Gershgorin, ExtractL/U(+diagonal), Scale*/AddScalar*, UpdateValuesCSR against the reference's host loops
    (host_matrix_csr.cpp:919-1160, :3465-3630) written out in Python: bit-exact"""
    from rocalution_amd import generators as gen
    rp, ci, va = gen.random_sparse(500, 7, seed=13)
    va = va.astype(dtype)
    # one row without a diagonal, one with a duplicated diagonal entry is not possible in sorted CSR: drop a diagonal
    keep = np.ones(len(ci), bool)
    i = 17
    keep[rp[i] + int(np.flatnonzero(ci[rp[i]:rp[i + 1]] == i)[0])] = False
    cnt = np.add.reduceat(keep.astype(np.int64), rp[:-1])
    ci, va = ci[keep], va[keep]
    rp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    n = len(rp) - 1
    rows = np.repeat(np.arange(n), np.diff(rp))
    A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va)
    # Gershgorin
    lo, hi = dtype(0), dtype(0)
    for r in range(n):
        s, d = dtype(0), dtype(0)
        for j in range(rp[r], rp[r + 1]):
            if ci[j] != r:
                s = dtype(s + abs(va[j]))
            else:
                d = va[j]
        hi = max(hi, dtype(s + d)); lo = min(lo, dtype(d - s))
    if _capi().has("ramd_mat_gershgorin"):  # (out of scope: only in a library built with -DRAMD_WITH_OFFSCOPE)
        glo, ghi = A.Gershgorin()
        assert (glo, ghi) == (float(lo), float(hi))
    # triangular parts
    for upper, diag in ((0, 0), (0, 1), (1, 0), (1, 1)):
        T = ra.LocalMatrix(dtype)
        (A.ExtractU if upper else A.ExtractL)(T, diag)
        m = (ci > rows) if (upper and not diag) else (ci >= rows) if upper else (ci <= rows) if diag else (ci < rows)
        trp, tci, tva = T.CopyToCSR()
        eq(trp, np.concatenate([[0], np.cumsum(np.bincount(rows[m], minlength=n))]).astype(np.int32))
        eq(tci, ci[m]); eq(tva, va[m])
    # value operations
    isd = ci == rows
    for name, which, op in (("Scale", None, "mul"), ("ScaleDiagonal", isd, "mul"), ("ScaleOffDiagonal", ~isd, "mul"),
                            ("AddScalar", None, "add"), ("AddScalarDiagonal", isd, "add"),
                            ("AddScalarOffDiagonal", ~isd, "add")):
        B = ra.LocalMatrix(dtype); B.SetDataPtrCSR(rp, ci, va)
        getattr(B, name)(0.3)
        exp = va.copy()
        sel = np.ones(len(va), bool) if which is None else which
        exp[sel] = (exp[sel] * dtype(0.3)) if op == "mul" else (exp[sel] + dtype(0.3))
        eq(B.CopyToCSR()[2], exp)
    B = ra.LocalMatrix(dtype); B.SetDataPtrCSR(rp, ci, va)
    B.UpdateValuesCSR(va[::-1].copy())
    eq(B.CopyToCSR()[2], va[::-1])


def test_coo_input_unsorted(ra, oracle):
    """SetDataPtrCOO with entries in arbitrary order: y = A x equals the reference's serial COO loop
    (y[row] += val * x[col] in storage order, host_matrix_coo.cpp:368-376) bit for bit"""
    rng = np.random.default_rng(23)
    n, nnz = 400, 3000
    row = rng.integers(0, n, nnz); col = rng.integers(0, n, nnz); val = rng.uniform(-1, 1, nnz)
    x = rng.uniform(-1, 1, n)
    A = ra.LocalMatrix(); A.SetDataPtrCOO(row, col, val, nrow=n, ncol=n)
    assert A.GetFormat() == ra.COO and A.GetNnz() == nnz
    y = ra.LocalVector(); y.Allocate("", n)
    A.Apply(ra.LocalVector(data=x), y)
    ref = np.zeros(n)
    for r, c, v in zip(row, col, val):
        ref[r] += v * x[c]
    eq(y.numpy(), ref)


def test_block_hyperplane_schedule_gives_identical_results():
    """RAMD_BLOCKSCHED_MIN=1 applies the hyperplane block order (blocksched.hip) to every level sweep and
    colouring of the kernel / solver tests: colours, permutations, ILU(0) factors and triangular solves must
    not change by a bit (also on patterns whose dependency sets overflow the per-block table)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RAMD_BLOCKSCHED_MIN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_kernels.py"),
                        os.path.join(here, "test_gpu_solvers.py"), "-q", "-m", "gpu", "-x", "-k",
                        "multicoloring or ilu or trisolve or preconditioner_apply or mcsgs or vs_oracle_larger"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("count", [1, 3, 8, 13, 30])
def test_fused_multi_axpy_equals_the_addscale_sequence(ra, dtype, count):
    """ramd_fused_multi_axpy (the GMRES solution update in one pass per eight basis vectors): the same additions per
    element in the same order as `count` AddScale calls -> bit-identical"""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    n = 100003
    rng = np.random.default_rng(count)
    x0 = rng.uniform(-1, 1, n).astype(dtype)
    vs = [ra.LocalVector(dtype, data=rng.uniform(-1, 1, n).astype(dtype)) for _ in range(count)]
    coef = rng.uniform(-2, 2, count)
    a = ra.LocalVector(dtype, data=x0)
    for v, c in zip(vs, coef):
        a.AddScale(v, float(dtype(c)))
    b = ra.LocalVector(dtype, data=x0)
    hs = (capi.vec_t * count)(*[v._h for v in vs])
    cs = (C.c_double * count)(*[float(dtype(c)) for c in coef])
    capi.check(lib.ramd_fused_multi_axpy(b._h, hs, cs, count))
    assert np.array_equal(a.numpy(), b.numpy())


def _mgs_sequential(lib, capi, w, vs, m):
    """the one-projection-per-pass path: multi_dot + m mgs_step launches; returns h[0..m-1], <w,w>"""
    first = (capi.vec_t * 1)(vs[0]._h)
    capi.check(lib.ramd_fused_multi_dot(first, 1, w._h, 0))
    for k in range(m):
        capi.check(lib.ramd_fused_mgs_step(w._h, vs[k]._h, k, vs[k + 1]._h if k + 1 < m else None, k + 1))
    out = np.zeros(m + 1)
    capi.check(lib.ramd_scalars_fetch(out.ctypes.data_as(capi.pf64), 0, m + 1))
    return out


def _mgs_blocked(lib, capi, w, vs, m):
    """ramd_fused_mgs_block exactly as GMRES::doFusedMGS drives it (include/rocalution/solvers.hpp)"""
    K = lib.ramd_fused_mgs_block_max()
    nsum = K + K * (K - 1) // 2
    areas = (124 - 2 * nsum, 124 - nsum)
    hs = (capi.vec_t * m)(*[v._h for v in vs[:m]])
    import ctypes as C
    at = lambda k: C.cast(C.byref(hs, k * C.sizeof(capi.vec_t)), C.POINTER(capi.vec_t))
    nb = (m + K - 1) // K
    for b in range(nb):
        nc = min(K, m - b * K)
        capi.check(lib.ramd_fused_mgs_block(w._h, at((b - 1) * K) if b else None, K if b else 0, (b - 1) * K,
                                            areas[(b - 1) & 1], at(b * K), nc, areas[b & 1]))
    nl = m - (nb - 1) * K
    capi.check(lib.ramd_fused_mgs_block(w._h, at((nb - 1) * K), nl, (nb - 1) * K, areas[(nb - 1) & 1], None, 0, m))
    out = np.zeros(m + 1)
    capi.check(lib.ramd_scalars_fetch(out.ctypes.data_as(capi.pf64), 0, m + 1))
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("m", [1, 2, 4, 5, 8, 9, 11, 16, 17, 30])
@pytest.mark.parametrize("basis", ["orthonormal", "oblique"])
def test_fused_mgs_block_is_the_mgs_recurrence(ra, dtype, m, basis):
    """ramd_fused_mgs_block (up to eight projections per pass; GMRES Arnoldi, gmres.cpp:480-486) against the one-projection-
    per-pass kernels and a float64 numpy MGS.  `oblique`: basis vectors far from orthogonal -- the block form measures
    their Gram entries, it does not assume an orthonormal basis.  Tolerance: the h of the two forms differ by rounding
    only, a few ulp of ||w|| (1e-13 / 2e-5 relative to ||w|| for fp64 / fp32, fp32 sums being accumulated in fp64)."""
    from rocalution_amd import capi
    lib = capi.load()
    n = 100003
    rng = np.random.default_rng(100 * m + (basis == "oblique"))
    V = rng.uniform(-1, 1, (n, m))
    if basis == "orthonormal":
        V, _ = np.linalg.qr(V)
    else:
        V /= np.linalg.norm(V, axis=0)
    w0 = rng.uniform(-1, 1, n)
    vs = [ra.LocalVector(dtype, data=np.ascontiguousarray(V[:, k]).astype(dtype)) for k in range(m)]
    wa = ra.LocalVector(dtype, data=w0.astype(dtype))
    wb = ra.LocalVector(dtype, data=w0.astype(dtype))
    ha = _mgs_sequential(lib, capi, wa, vs, m)
    hb = _mgs_blocked(lib, capi, wb, vs, m)
    # numpy MGS on the rounded inputs
    wr = w0.astype(dtype).astype(np.float64)
    href = np.zeros(m + 1)
    for k in range(m):
        vk = vs[k].numpy().astype(np.float64)
        href[k] = vk @ wr
        wr = wr - href[k] * vk
    href[m] = wr @ wr
    scale = np.linalg.norm(w0)
    tol = (1e-13 if dtype == np.float64 else 2e-5) * (1.0 if basis == "orthonormal" else 20.0)
    assert np.max(np.abs(hb[:m] - ha[:m])) <= tol * scale, (hb, ha)
    assert np.max(np.abs(hb[:m] - href[:m])) <= tol * scale
    assert abs(hb[m] - ha[m]) <= 4 * tol * scale * scale and abs(hb[m] - href[m]) <= 4 * tol * scale * scale
    assert np.max(np.abs(wb.numpy().astype(np.float64) - wa.numpy().astype(np.float64))) <= tol * scale
    assert np.max(np.abs(wb.numpy().astype(np.float64) - wr)) <= tol * scale


def test_fused_mgs_block_rejects_bad_arguments(ra):
    from rocalution_amd import capi
    lib = capi.load()
    K = lib.ramd_fused_mgs_block_max()
    v = [ra.LocalVector(np.float64, data=np.ones(64)) for _ in range(K + 1)]
    w = ra.LocalVector(np.float64, data=np.ones(64))
    hs = (capi.vec_t * (K + 1))(*[x._h for x in v])
    assert lib.ramd_fused_mgs_block(w._h, None, 0, 0, 0, None, 0, 0) != 0           # nothing to do
    assert lib.ramd_fused_mgs_block(w._h, None, 0, 0, 0, hs, K + 1, 0) != 0         # block too long
    assert lib.ramd_fused_mgs_block(w._h, hs, K - 1, 0, 40, hs, 2, 90) != 0         # a followed block must be full
    assert lib.ramd_fused_mgs_block(w._h, hs, K, 0, 40, hs, K, 44) != 0             # overlapping slot areas
    assert lib.ramd_fused_mgs_block(w._h, None, 0, 0, 0, hs, K, 504) != 0           # sums beyond the record (512 slots)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 255, 257, 1023])
def test_fused_mgs_block_tiny_vectors(ra, n):
    """vectors shorter than a 16-byte packet / a workgroup: only the scalar tail loop (or a partial packet pass) runs"""
    from rocalution_amd import capi
    lib = capi.load()
    rng = np.random.default_rng(n)
    for dtype in (np.float64, np.float32):
        m = min(n, 6)
        V = rng.uniform(-1, 1, (n, m))
        V /= np.linalg.norm(V, axis=0)
        w0 = rng.uniform(-1, 1, n)
        vs = [ra.LocalVector(dtype, data=np.ascontiguousarray(V[:, k]).astype(dtype)) for k in range(m)]
        wa = ra.LocalVector(dtype, data=w0.astype(dtype))
        wb = ra.LocalVector(dtype, data=w0.astype(dtype))
        ha = _mgs_sequential(lib, capi, wa, vs, m)
        hb = _mgs_blocked(lib, capi, wb, vs, m)
        tol = (1e-13 if dtype == np.float64 else 2e-5) * 20.0 * max(1.0, np.linalg.norm(w0))
        assert np.max(np.abs(hb[:m] - ha[:m])) <= tol
        assert abs(hb[m] - ha[m]) <= 4 * tol * max(1.0, np.linalg.norm(w0))
        assert np.max(np.abs(wb.numpy().astype(np.float64) - wa.numpy().astype(np.float64))) <= tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_csr_row_patterns_in_a_fresh_process(dtype):
    """row-pattern SpMV (csr_analyse_pattern): a 3-D stencil falls into <= 27 patterns and takes the dictionary kernel, a
    random matrix and a matrix with a 29-entry row do not (patterns hold up to 28 entries since round 6: the 27 of the reference's own 3-D operator); results bit-exact against the oracle either way, also after the
    values were replaced and for a rectangular block"""
    import os
    import subprocess
    import sys
    code = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import rocalution_amd as ra
from rocalution_amd import capi, generators as gen
from oracle import oracle
oracle.build(); oracle.set_threads(1); lib = capi.load(); ra.init_rocalution()
dtype = np.%s
def info(A):
    s, n, w = C.c_int(9), C.c_int(0), C.c_int(0)
    capi.check(lib.ramd_mat_pattern_info(A._h, C.byref(s), C.byref(n), C.byref(w))); return s.value, n.value, w.value
def check(rp, ci, va, nrow, ncol, want):
    rng = np.random.default_rng(5)
    A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va.astype(dtype), nrow=nrow, ncol=ncol)
    assert info(A)[0] == 0
    xh = rng.uniform(-1, 1, ncol).astype(dtype); y0 = rng.uniform(-1, 1, nrow).astype(dtype)
    x = ra.LocalVector(dtype, data=xh); y = ra.LocalVector(dtype); y.Allocate("", nrow)
    A.Apply(x, y)
    st = info(A); assert st[0] == want, (st, want)
    assert np.array_equal(y.numpy(), oracle.csr_apply(rp, ci, va.astype(dtype), xh))
    ya = ra.LocalVector(dtype, data=y0); A.ApplyAdd(x, 0.375, ya)
    assert np.array_equal(ya.numpy(), oracle.csr_apply_add(rp, ci, va.astype(dtype), xh, dtype(0.375), y0))
    return A, st
rp, ci, va = gen.poisson7(21, np.float64)
n = len(rp) - 1
va = va * np.random.default_rng(1).uniform(0.5, 1.5, len(va))       # values play no role in the pattern
A, st = check(rp, ci, va, n, n, 1)
assert st[1] <= 27 and st[2] == 28, st
# the same operator as ELL and as HYB: the analysis runs on the slot tuples of the ELL block (empty slots included)
for fmt in (ra.ELL, ra.HYB):
    B = ra.LocalMatrix(dtype); B.SetDataPtrCSR(rp, ci, va.astype(dtype))
    assert B.ConvertTo(fmt) == fmt and info(B)[0] == 0
    xh = np.random.default_rng(6).uniform(-1, 1, n).astype(dtype); y0 = np.random.default_rng(7).uniform(-1, 1, n).astype(dtype)
    x = ra.LocalVector(dtype, data=xh); y = ra.LocalVector(dtype); y.Allocate("", n)
    B.Apply(x, y)
    assert info(B)[0] == 1 and info(B)[1] <= 27, info(B)
    assert np.array_equal(y.numpy(), oracle.csr_apply(rp, ci, va.astype(dtype), xh))
    ya = ra.LocalVector(dtype, data=y0); B.ApplyAdd(x, -1.25, ya)
    assert np.array_equal(ya.numpy(), oracle.csr_apply_add(rp, ci, va.astype(dtype), xh, dtype(-1.25), y0))
# a rectangular row block of the same operator (columns keep their global numbering): offsets relative to the local row
r0, r1 = 3 * 441, 9 * 441
rpb = (rp[r0:r1 + 1] - rp[r0]).astype(np.int32); cib = ci[rp[r0]:rp[r1]]; vab = va[rp[r0]:rp[r1]]
check(rpb, cib, vab, r1 - r0, n, 1)
# random pattern: far more than 64 different rows
rng = np.random.default_rng(2); m = 3000
rows = [sorted(set(rng.integers(0, m, 5).tolist())) for _ in range(m)]
rpr = np.zeros(m + 1, np.int32); rpr[1:] = np.cumsum([len(r) for r in rows])
cir = np.array([c for r in rows for c in r], np.int32); var = rng.uniform(-1, 1, len(cir))
check(rpr, cir, var, m, m, -1)
# one row of 29 entries in an otherwise structured matrix
rows = [[i] for i in range(m)]; rows[77] = list(range(48, 77)) ; rows[78] = []
rpl = np.zeros(m + 1, np.int32); rpl[1:] = np.cumsum([len(r) for r in rows])
cil = np.array([c for r in rows for c in r], np.int32); val = rng.uniform(-1, 1, len(cil))
check(rpl, cil, val, m, m, -1)
rows[77] = list(range(49, 77))                                         # 28 entries: fits; an empty row is a pattern too
rpl = np.zeros(m + 1, np.int32); rpl[1:] = np.cumsum([len(r) for r in rows])
cil = np.array([c for r in rows for c in r], np.int32); val = rng.uniform(-1, 1, len(cil))
A2, st = check(rpl, cil, val, m, m, 1)
assert st[1] == 3 and st[2] == 28, st
print("OK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)), np.dtype(dtype).name)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RAMD_CSR_PAT="1"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0 and b"OK" in r.stdout, r.stdout.decode()[-3000:]


def test_csr_row_patterns_give_up_quickly_on_a_large_unstructured_matrix(ra, oracle):
    """2^20 rows of four random columns each (4.2 M entries: above the default threshold of the analysis): thousands of
    distinct rows, so the pattern sweep must stop after the first few thousand rows instead of probing a full table for
    every row; the product is the ordinary one, bit-exact"""
    import ctypes as C
    import time
    from rocalution_amd import capi
    lib = capi.load()
    n = 1 << 20
    rng = np.random.default_rng(8)
    ci = np.sort(rng.integers(0, n, (n, 4)), axis=1).astype(np.int32).ravel()
    rp = (np.arange(n + 1, dtype=np.int64) * 4).astype(np.int32)
    va = rng.uniform(-1, 1, 4 * n)
    xh = rng.uniform(-1, 1, n)
    A = ra.LocalMatrix(np.float64); A.SetDataPtrCSR(rp, ci, va)
    x = ra.LocalVector(np.float64, data=xh); y = ra.LocalVector(np.float64); y.Allocate("", n)
    ra.sync()
    t0 = time.perf_counter()
    A.Apply(x, y)
    ra.sync()
    dt = time.perf_counter() - t0
    st = C.c_int(9)
    capi.check(lib.ramd_mat_pattern_info(A._h, C.byref(st), None, None))
    if os.environ.get("RAMD_CSR_PAT", "-1") != "0":
        assert st.value == -1
    assert dt < 1.0, dt
    eq(y.numpy(), oracle.csr_apply(rp, ci, va, xh))


# (three switches of the colour sweeps ride along with SpMV variants they have no kernel in common with -- RAMD_MC_RB=2 with the
#  four-lane product, RAMD_MC_FOLD=0 with the LDS-staged x pieces, RAMD_MC_RB=0 with the two-block columns product: three
#  interpreter starts less)
SPMV_VARIANTS = ["RAMD_CSR_Q4=1,RAMD_MC_RB=2", "RAMD_CSR_PAT=1", "RAMD_CSR_PAT=1,RAMD_CSR_XL=1,RAMD_MC_FOLD=0", "RAMD_CSR_PAT=0",
                 "RAMD_CSR_PAT=0,RAMD_CSR_GRP=1", "RAMD_CSR_PAT=1,RAMD_CSR_PAT2=0", "RAMD_ELL2=1,RAMD_CSR_PAT=1", "RAMD_ELL2=1,RAMD_CSR_PAT=0",
                 "RAMD_CSR_PAT=0,RAMD_CSR_COL2=2,RAMD_MC_RB=0", "RAMD_CSR_PAT=1,RAMD_CSR_NORP=1", "RAMD_CSR_PAT=0,RAMD_CSR_W4=1",
                 "RAMD_CSR_PAT=0,RAMD_CSR_W4=1,RAMD_CSR_W4_WAVES=1", "RAMD_CSR_PAT=0,RAMD_CSR_W4=1,RAMD_CSR_WP=0", "RAMD_CSR_PAT=0,RAMD_CSR_PIPE=1",
                 "RAMD_CSR_PAT=0,RAMD_CSR_W4=0"]


def _spmv_family():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    jobs = {}
    for variant in SPMV_VARIANTS:
        env = dict(os.environ)
        for kv in variant.split(","):
            env[kv.split("=")[0]] = kv.split("=")[1]
        cmd = [sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_kernels.py"), os.path.join(here, "test_gpu_edge_cases.py"),
               os.path.join(here, "test_gpu_solvers.py"), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider", "-k",
               "(spmv or csr or apply or fused_bicgstab or golden or smoother or convert or mcsgs or mcgs or mcilu or multicolor or history) "
               "and not fresh_process"]
        jobs[variant] = (cmd, env, 1500)
    return jobs


@pytest.mark.parametrize("variant", SPMV_VARIANTS)
def test_spmv_variants_forced_in_a_fresh_process(variant):
    """the CSR SpMV has an opt-in four-lanes-per-row walk (k_csr_q4), and CSR / ELL / HYB products and the multi-colour
    sweeps rebuild the columns of structured matrices from row patterns (by default only from 2^20 entries on; with RAMD_CSR_XL=1
    the CSR product stages the x pieces its 256-row blocks need in LDS, k_csr_xl, instead of gathering x; RAMD_CSR_W4=1: wave-private passes -- products formed where the packets land, k_csr_wp, or (RAMD_CSR_WP=0) four lanes per row, k_csr_w4; RAMD_CSR_W4=0: rows of 16+ entries (the shell surrogates) in the wave-private row walk the stencils take, k_csr_wr, several passes per wave; RAMD_CSR_PIPE=1: the next
    pass requested before the row walk; with row patterns the SGS sweeps fold colour 0 into its readers, RAMD_MC_FOLD=0: do not; RAMD_MC_RB=2: the one-pass red-black lattice form of the SGS apply, k_mc_rb, on every two-colour lattice operator however small, =0: never); each forced
    on (or off) for EVERY matrix of the SpMV / ApplyAdd / fused-dot / Jacobi-sweep / format / multi-colour / solver-history
    tests: results must not change (bit-exact: same values, same order).  The sixteen processes run several at a time
    (conftest.forced_run)."""
    from conftest import forced_run
    rc, out = forced_run("spmv", variant, _spmv_family())
    assert rc == 0, out[-3000:]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_scalar_programs_and_guarded_combines(ra, dtype):
    """The device-resident scalar algebra of the recurrence-based drivers (ramd_scalars_eval, ramd_vec_combine_s):
    a program on the record gives what the host would compute with the same operations in the same order (IEEE double, or
    float for float drivers: hip_vector.cpp returns scalars to the host, the drivers of src/solvers/krylov do this
    arithmetic there); a combine is the reference's AddScale / ScaleAdd / ScaleAdd2 expression with the coefficient read
    from a slot (bit-exact against the kernels taking the coefficient by value); a raised flag turns later combines into
    no-ops."""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()

    class Sop(C.Structure):
        _fields_ = [("op", C.c_int), ("dst", C.c_int), ("a", C.c_int), ("b", C.c_int), ("imm", C.c_double)]
    SET, MOV, ADD, SUB, MUL, DIV, NEG, SQRT, ABS, ZFLAG = range(10)
    single = 1 if dtype == np.float32 else 0
    T = np.float32 if single else np.float64
    base = 200
    prog = [(SET, 0, -1, -1, 3.7), (SET, 1, -1, -1, -1.3), (DIV, 2, 0, 1, 0), (MUL, 3, 2, 2, 0), (SET, 4, -1, -1, 1.0),
            (ADD, 5, 4, 3, 0), (SQRT, 5, 5, -1, 0), (DIV, 6, 4, 5, 0), (NEG, 7, 6, -1, 0), (SUB, 8, 7, 0, 0), (ABS, 9, 8, -1, 0),
            (MOV, 10, 9, -1, 0), (SET, 11, -1, -1, 0.0), (ZFLAG, 11, 10, -1, 0), (SET, 12, -1, -1, 0.0), (ZFLAG, 13, 12, -1, 0)]
    capi.check(lib.ramd_scalars_set(base + 13, 0.0))
    arr = (Sop * len(prog))(*[Sop(o, base + d, (base + a) if a >= 0 else -1, (base + b) if b >= 0 else -1, imm) for o, d, a, b, imm in prog])
    capi.check(lib.ramd_scalars_eval(arr, len(prog), single))
    out = (C.c_double * 14)()
    capi.check(lib.ramd_scalars_fetch(out, base, 14))
    s = [T(0)] * 14
    s[0], s[1] = T(3.7), T(-1.3)
    s[2] = T(s[0] / s[1]); s[3] = T(s[2] * s[2]); s[4] = T(1.0); s[5] = T(np.sqrt(T(s[4] + s[3]))); s[6] = T(s[4] / s[5])
    s[7] = T(-s[6]); s[8] = T(s[7] - s[0]); s[9] = T(abs(s[8])); s[10] = s[9]
    assert [out[k] for k in range(11)] == [float(v) for v in s[:11]]
    assert out[11] == 0.0 and out[13] == 1.0  # flag only where the tested slot is zero
    # combines: coefficient from a slot == coefficient by value
    n = 100003
    rng = np.random.default_rng(11)
    hx, hy, hz = (rng.uniform(-1, 1, n).astype(dtype) for _ in range(3))
    coef = float(out[2])
    V = lambda h: ra.LocalVector(dtype, data=h)

    def combine(x, terms, guard=-1):
        vs = (capi.vec_t * len(terms))(*[t[0]._h for t in terms])
        sl = (C.c_int * len(terms))(*[t[1] for t in terms])
        fa = (C.c_double * len(terms))(*[t[2] for t in terms])
        capi.check(lib.ramd_vec_combine_s(x._h, len(terms), vs, sl, fa, guard))
    x, y, z = V(hx), V(hy), V(hz)
    ref = V(hx); ref.AddScale(y, -coef)
    combine(x, [(x, -1, 1.0), (y, base + 2, -1.0)])
    assert np.array_equal(x.numpy(), ref.numpy())
    x = V(hx); ref = V(hx); ref.ScaleAdd(coef, y)
    combine(x, [(x, base + 2, 1.0), (y, -1, 1.0)])
    assert np.array_equal(x.numpy(), ref.numpy())
    x = V(hx); ref = V(hx); ref.ScaleAdd2(coef, y, float(out[6]), z, -1.0)
    combine(x, [(x, base + 2, 1.0), (y, base + 6, 1.0), (z, -1, -1.0)])
    assert np.array_equal(x.numpy(), ref.numpy())
    x = V(hx); ref = V(hx); ref.Scale(coef)
    combine(x, [(x, base + 2, 1.0)])
    assert np.array_equal(x.numpy(), ref.numpy())
    # guarded: slot base+13 holds 1 -> nothing happens; slot base+11 holds 0 -> the update runs
    x = V(hx)
    combine(x, [(x, -1, 1.0), (y, base + 2, 1.0)], guard=base + 13)
    assert np.array_equal(x.numpy(), hx)
    combine(x, [(x, -1, 1.0), (y, base + 2, 1.0)], guard=base + 11)
    ref = V(hx); ref.AddScale(y, coef)
    assert np.array_equal(x.numpy(), ref.numpy())


@pytest.mark.gpu
def test_single_launch_reductions_under_load(ra):
    """The single-launch reductions hand their per-workgroup partial sums to the last workgroup through relaxed agent-scope
    atomics, a store-completion wait and a ticket -- no release fence (device_utils.hpp: grid_reduce_finish; a fence there
    is a write-back of an XCD's L2).  Stress: the largest grids the kernels use (8192 workgroups spread over all eight
    XCDs), hundreds of launches back to back, with sums whose exact value is known: a partial that arrived late or stale
    would show up as a wrong integer."""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    n = 8192 * 256 * 4 + 37
    rng = np.random.default_rng(3)
    hx = rng.integers(-3, 4, n).astype(np.float64)
    hy = rng.integers(-3, 4, n).astype(np.float64)
    x, y = ra.LocalVector(data=hx), ra.LocalVector(data=hy)
    exact_dot, exact_nrm2 = float(np.dot(hx, hy)), float(np.dot(hx, hx))
    for rep in range(150):
        assert x.Dot(y) == exact_dot, rep
        assert x.Norm() ** 2 == pytest.approx(exact_nrm2, rel=1e-15), rep
    hs = (capi.vec_t * 2)(x._h, y._h)
    out = (C.c_double * 2)()
    for rep in range(150):
        capi.check(lib.ramd_fused_multi_dot(hs, 2, y._h, 40))
        capi.check(lib.ramd_scalars_fetch(out, 40, 2))
        assert (out[0], out[1]) == (exact_dot, float(np.dot(hy, hy))), rep


def test_placement_entry_points_keep_contents_and_count_trials(ra):
    """ramd_vec_allocate_apart / ramd_vec_place_apart / ramd_vec_place_by_trial / ramd_vec_placement_class move a vector
    between device blocks and never change what it holds; the trial callback runs 3 x (1 + tries) times when no stop ratio
    is given (a warm run and two timed ones per placement -- ranks that exchange halos inside it rely on the count), fewer
    with one; vectors below 64 MiB are left alone; ramd_placement_seconds accounts for the time."""
    import ctypes as C
    from rocalution_amd import capi
    lib = capi.load()
    n = (96 << 20) // 8  # 96 MiB of fp64: above the 64-MiB threshold
    rng = np.random.default_rng(11)
    a = rng.uniform(-1, 1, n)
    b = rng.uniform(-1, 1, n)
    va = ra.LocalVector(data=a)
    vz = ra.LocalVector()
    capi.check(lib.ramd_vec_allocate_apart(vz._h, n, va._h))
    assert vz.GetSize() == n and not vz.numpy().any()  # (zero-filled like Allocate)
    vb = ra.LocalVector(data=b)
    cls = C.c_int(-7)
    capi.check(lib.ramd_vec_placement_class(va._h, C.byref(cls)))
    assert cls.value in (0, 1)
    capi.check(lib.ramd_placement_seconds(None, 1))
    moved = C.c_int(0)
    capi.check(lib.ramd_vec_place_apart(vb._h, va._h, C.byref(moved)))
    assert np.array_equal(va.numpy(), a) and np.array_equal(vb.numpy(), b)
    calls = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p)

    def run(_ctx):
        calls.append(1)
        vb.Scale(1.0)  # some kernel that touches the vector
        return 0
    cb_keep = CB(run)
    cb = C.cast(cb_keep, C.c_void_p)
    capi.check(lib.ramd_vec_place_by_trial(vb._h, cb, None, 2, 0.0, None, C.byref(moved)))
    assert len(calls) == 3 * (1 + 2)
    assert np.array_equal(vb.numpy(), b)
    del calls[:]
    capi.check(lib.ramd_vec_place_by_trial(vb._h, cb, None, 6, 0.5, va._h, C.byref(moved)))  # (no block is twice as fast)
    assert len(calls) == 3 * (1 + 6) and np.array_equal(vb.numpy(), b)
    secs = C.c_double(0.0)
    capi.check(lib.ramd_placement_seconds(C.byref(secs), 0))
    assert secs.value > 0.0
    small = ra.LocalVector(data=np.arange(1000.0))
    del calls[:]
    capi.check(lib.ramd_vec_place_by_trial(small._h, cb, None, 3, 0.0, None, C.byref(moved)))
    assert len(calls) == 0 and moved.value == 0 and np.array_equal(small.numpy(), np.arange(1000.0))
    capi.check(lib.ramd_vec_placement_class(small._h, C.byref(cls)))
    assert cls.value == -1
