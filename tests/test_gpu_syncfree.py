"""The sync-free grouped triangular solve (k_trsv_sf, trisolve.hip): the form a deep, narrow dependency graph of long rows takes
when the tile coordinates find no chains -- a shell mesh (5 unknowns per node) in reverse Cuthill-McKee order.

By default only matrices of >= 4096 rows with fewer than 2048 rows per group level take it; RAMD_TRSV_SF=2 forces it on every
matrix whose rows have at most 48 entries outside their row group.  The settings are read once per process, hence the
subprocesses.  Everything is compared bit for bit: the kernel performs the operations of host_matrix_csr.cpp:1163-1221 (LUSolve),
:1357-1404 (LSolve), :1420-1466 (USolve) per row in their order."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = ("((ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0) "
          "and not full_size and not cpp and not fresh_process) or variants_of_the_class")
FORCED = dict(RAMD_TRSV_SF="2", RAMD_TRSV_CT="0", RAMD_TRSV_LAT="0", RAMD_TRSV_CT_VERBOSE="1")


def forced_job():
    env = dict(os.environ, **FORCED)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_kernels.py"), os.path.join(ROOT, "tests", "test_gpu_solvers.py"),
           os.path.join(ROOT, "tests", "test_gpu_shell.py"), "-k", SELECT, "--durations=5"]
    return cmd, env, 1500


@pytest.mark.gpu
def test_parity_suite_with_the_sync_free_grouped_form_forced():
    """ILU(0) / IC factors + LUSolve / LLSolve / LSolve / USolve goldens, preconditioner applies, solver histories, the config-3
    class in four numberings (factors, LUSolve and GMRES(30)+ILU(0) against the oracle) -- with every triangular plan that can be
    in the sync-free grouped form (also the descending-order sweep of LLSolve's second stage)."""
    from conftest import forced_run
    from test_gpu_box_tiles_forced import _tri_family
    rc, out = forced_run("tri", "syncfree", _tri_family())
    tail = out[-3000:]
    assert rc == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    assert "sync-free grouped plan (lower)" in out and "sync-free grouped plan (upper)" in out, tail
    assert "8 lanes per row" in out and "4 lanes per row" in out, tail


_BLOCKS = r"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, %(root)r)
import rocalution_amd as ra
from rocalution_amd import solvers as S
from oracle import oracle
ra.init_rocalution()
rng = np.random.default_rng(23)


def block_matrix(nn, bmax, deg, reach, symmetric):
    # a node graph (node i linked to up to `deg` nodes within `reach` before it) expanded with b_i unknowns per node and dense
    # coupling blocks: supernodes of 1 .. bmax rows (runs of more than 8 rows are cut), rows of very different lengths; an
    # unsymmetric pattern gives the two triangles different row groups
    b = rng.integers(1, bmax + 1, nn)
    off = np.concatenate([[0], np.cumsum(b)])
    n = int(off[-1])
    rows, cols = [], []
    for i in range(nn):
        js = set(int(j) for j in rng.integers(max(0, i - reach), i + 1, size=int(rng.integers(0, deg + 1)))) | {i}
        for j in js:
            for a in range(off[i], off[i + 1]):
                for c in range(off[j], off[j + 1]):
                    rows.append(a); cols.append(c)
                    if symmetric or rng.random() < 0.5:
                        rows.append(c); cols.append(a)
    v = rng.uniform(-1, -0.05, len(rows))
    A = sp.coo_matrix((v, (rows, cols)), shape=(n, n)).tocsr(); A.sum_duplicates()
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + np.asarray(abs(A).sum(axis=0)).ravel() + 1.0)
    A = A.tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


checked = 0
for nn, bmax, deg, reach, sym in ((1, 1, 0, 1, True), (1, 9, 0, 1, True), (40, 5, 3, 6, True), (700, 5, 4, 30, True), (900, 9, 3, 12, False),
                                  (400, 12, 2, 5, True), (1500, 3, 6, 200, False), (600, 8, 5, 9, True)):
    rp, ci, va = block_matrix(nn, bmax, deg, reach, sym)
    n = len(rp) - 1
    for dt in (np.float64, np.float32):
        v = va.astype(dt)
        A = ra.LocalMatrix(dt); A.SetDataPtrCSR(rp, ci, v)
        A.ILU0Factorize()
        lu = oracle.ilu0(rp, ci, v)
        assert np.array_equal(A.CopyToCSR()[2], lu), ("ilu0", nn, bmax, dt)
        A.LUAnalyse()
        y = ra.LocalVector(dt); y.Allocate("", n)
        for rep in range(2):
            b = rng.uniform(-1, 1, n).astype(dt)
            A.LUSolve(ra.LocalVector(dt, data=b), y)
            assert np.array_equal(y.numpy(), oracle.lusolve(rp, ci, lu, b)), ("lusolve", nn, bmax, dt)
        for unit in (False, True):
            B = ra.LocalMatrix(dt); B.SetDataPtrCSR(rp, ci, v)
            B.LAnalyse(unit); B.LSolve(ra.LocalVector(dt, data=b), y)
            assert np.array_equal(y.numpy(), oracle.lsolve(rp, ci, v, b, unit)), ("lsolve", nn, bmax, dt, unit)
            B.UAnalyse(unit); B.USolve(ra.LocalVector(dt, data=b), y)
            assert np.array_equal(y.numpy(), oracle.usolve(rp, ci, v, b, unit)), ("usolve", nn, bmax, dt, unit)
        if sym:  # IC: L L^T solve, the second stage takes a row's entries in descending order (host_matrix_csr.cpp:1294-1341)
            A0 = ra.LocalMatrix(dt); A0.SetDataPtrCSR(rp, ci, v)
            ls = S.CG(dt); ls.SetOperator(A0); ls.SetPreconditioner(S.IC()); ls.Build()
            z = ra.LocalVector(dt); z.Allocate("", n)
            ls.PrecondApply(ra.LocalVector(dt, data=b), z)
            assert np.array_equal(z.numpy(), oracle.precond_apply(oracle.PC_IC, rp, ci, v, b)), ("ic", nn, bmax, dt)
            ls.Clear()
    print("ok nodes=%%d n=%%d unknowns per node <= %%d" %% (nn, n, bmax), flush=True)
    checked += 1
print("checked", checked)
"""


@pytest.mark.gpu
def test_row_groups_of_1_to_12_rows_unsymmetric_patterns_fp64_fp32_bit_exact(tmp_path):
    """Block matrices with 1 .. 12 unknowns per node (row groups of every size, runs cut at 8 rows), symmetric and unsymmetric
    patterns (the upper triangle's groups differ from the lower one's), rows of a few to beyond 48 entries (those matrices fall
    back, loudly), n = 1 and a single group: ILU(0), LUSolve, LSolve / USolve with stored and with unit diagonal, the IC apply
    (L L^T solve: the second stage in descending order) against the oracle, fp64 and fp32."""
    script = tmp_path / "blocks.py"
    script.write_text(_BLOCKS % {"root": ROOT})
    p = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=dict(os.environ, **FORCED), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "checked 8" in p.stdout, p.stdout[-4000:]
    assert "sync-free grouped plan (lower)" in p.stdout and "sync-free grouped plan (upper)" in p.stdout, p.stdout[-2000:]
    assert "row groups of <= 8 rows" in p.stdout and "row groups of <= 5 rows" in p.stdout, p.stdout[-2000:]


@pytest.mark.gpu
def test_full_size_rcm_shell_takes_the_form_by_default():
    """The config-3 class at full size (n = 1 507 005) in reverse Cuthill-McKee order: the tiles give up, both triangles take the
    sync-free grouped form (8 lanes per row, groups of 5 rows = the mesh nodes, ~2 140 group levels instead of 10 700 row levels),
    L U x = b holds to rounding, twenty solves in a row are bit-identical, GMRES(30)+ILU(0) converges to the known solution."""
    import ctypes as C
    import numpy as np
    import rocalution_amd as ra
    from rocalution_amd import capi, generators as gen, solvers as S
    ra.init_rocalution()
    lib = capi.load()
    rp, ci, va = gen.shell_variant(549, "rcm")
    n = len(rp) - 1
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    F = ra.LocalMatrix(); F.CloneFrom(A)
    F.ILU0Factorize(); F.LUAnalyse()
    st = (C.c_int64 * 16)()
    default = os.environ.get("RAMD_TRSV_SF", "1") == "1" and os.environ.get("RAMD_TRSV_CT", "1") != "0"
    for which in (0, 1):
        capi.check(lib.ramd_tri_plan_stats(which, st))
        if default:
            assert st[0] == 6 and st[12] != 0, list(st)  # the sync-free grouped form, and why the tiles were not taken
            assert st[8] == 8 and 5 <= st[9] <= 8 and abs(st[4] - n // 5) <= 8, list(st)  # (a few runs of rows reach over two nodes)
            assert 2000 <= st[2] <= 2300, list(st)
    rng = np.random.default_rng(3)
    bh = rng.uniform(-1.0, 1.0, n)
    b = ra.LocalVector(data=bh)
    x = ra.LocalVector(); x.Allocate("", n)
    F.LUSolve(b, x)
    xh = x.numpy().copy()
    frp, fci, fva = F.CopyToCSR()
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    upper = fci >= rows
    u = np.bincount(rows[upper], weights=fva[upper] * xh[fci[upper]], minlength=n)
    lower = ~upper
    bl = u + np.bincount(rows[lower], weights=fva[lower] * u[fci[lower]], minlength=n)
    assert np.max(np.abs(bl - bh)) <= 1e-11 * max(1.0, np.max(np.abs(u)))
    for rep in range(20):
        F.LUSolve(b, x)
        assert np.array_equal(x.numpy(), xh), rep
    ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
    rhs = ra.LocalVector(); rhs.Allocate("", n)
    A.Apply(ones, rhs)
    x.Zeros()
    ls = S.GMRES(); ls.SetOperator(A); ls.SetPreconditioner(S.ILU()); ls.SetBasisSize(30)
    ls.Init(1e-15, 1e-6, 1e8, 2000); ls.Build()
    ls.Solve(rhs, x)
    assert ls.GetSolverStatus() == 2 and 40 <= ls.GetIterationCount() <= 160, (ls.GetIterationCount(), ls.GetSolverStatus())
    x.AddScale(ones, -1.0)
    assert x.Norm() / np.sqrt(n) < 1e-4
    ls.Clear()


@pytest.mark.gpu
def test_bench_mtx_reports_the_plan_of_a_file_nobody_here_has_seen(tmp_path):
    """`bench.py --mtx PATH` is what a holder of SuiteSparse af_shell10.mtx would run (config 3; the file cannot be fetched here):
    a MatrixMarket `symmetric` file of the same class in reverse Cuthill-McKee order goes through ReadFileMTX, and the line comes
    back with the rate AND the diagnosis of its triangular solves -- the form each triangle took, its dependency levels, and why the
    tiles were not used."""
    import json
    from rocalution_amd import generators as gen
    rp, ci, va = gen.shell_variant(40, "rcm")  # 8000 rows
    path = str(tmp_path / "shell40_rcm.mtx")
    gen.write_mtx_symmetric(path, rp, ci, va)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mtx", path, "--solver", "gmres", "--precond", "ilu0", "--steps", "12",
           "--warmup", "3", "--no-cpu-baseline", "--no-reference-gpu"]
    p = subprocess.run(cmd, cwd=str(tmp_path), env=dict(os.environ, TMPDIR=str(tmp_path)), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["steps"] == 12 and d["value"] > 0 and "shell40_rcm.mtx" in d["config"]["workload"], d["config"]
    assert d["ingest"]["file_bytes"] == os.path.getsize(path)
    tp = d["tri_plan"]
    for which in ("lower", "upper"):
        assert tp[which]["rows"] == len(rp) - 1 and tp[which]["dependency_levels"] > 0, tp[which]
        assert tp[which]["form"], tp[which]
        if "box tiles" not in tp[which]["form"] and "lattice" not in tp[which]["form"]:
            assert tp[which]["why_not_box_tiles"], tp[which]
    if os.environ.get("RAMD_TRSV_SF", "1") == "1" and os.environ.get("RAMD_TRSV_CT", "1") != "0":
        assert "k_trsv_sf" in tp["lower"]["form"] and "k_trsv_sf" in tp["upper"]["form"], tp
    assert d["roofline"]["kernel"].startswith("sparse triangular solve"), d["roofline"]["kernel"]


@pytest.mark.gpu
def test_short_division_sequence_gives_the_bits_of_the_division(tmp_path):
    """trsv_syncfree.hip sf_div_short / sf_quotient_in_window: inside its exponent window the solve forms a / d with the last three
    operations of gfx950's fp64 division sequence, the divisor-only part (reciprocal, two Newton steps) coming from the plan; whether
    the operands were inside is read off the QUOTIENT (one test per unit of rows instead of one per division).  Same instructions on the same
    operands => the bits of `/` (and of the host's correctly rounded division, which the reference performs:
    host_matrix_csr.cpp:1216).  Checked here on 12 M operand pairs: random mantissas over the whole window and across its edges,
    divisors with all-ones / all-zeros mantissas, equal operands, powers of two, zeros / infinities / NaNs / denormals (which must
    leave the window and take `/`), and every diagonal of the ILU(0) factors of the five numberings of the config-3 class against
    right-hand sides of the magnitudes a solve meets."""
    import ctypes as C
    import numpy as np
    import rocalution_amd as ra
    from rocalution_amd import capi, generators as gen
    ra.init_rocalution()
    lib = capi.load()
    rng = np.random.default_rng(11)

    def probe(a, d):
        n = len(a)
        va, vd = ra.LocalVector(data=a), ra.LocalVector(data=d)
        vf = ra.LocalVector(); vf.Allocate("", n)
        vp = ra.LocalVector(); vp.Allocate("", n)
        vi = ra.LocalVector(np.int32); vi.Allocate("", n)
        ptr = lambda v: C.c_void_p(lib.ramd_vec_data(v._h))
        capi.check(lib.ramd_selftest_sf_div(n, ptr(va), ptr(vd), ptr(vf), ptr(vp), ptr(vi)))
        return vf.numpy(), vp.numpy(), vi.numpy()

    def rand(n, elo, ehi):
        m = rng.integers(0, 1 << 52, n, dtype=np.uint64)
        e = rng.integers(elo, ehi + 1, n, dtype=np.uint64)
        s = rng.integers(0, 2, n, dtype=np.uint64)
        return ((s << np.uint64(63)) | (e << np.uint64(52)) | m).view(np.float64)

    n = 4_000_000
    cases = []
    cases.append((rand(n, 640, 1407), rand(n, 640, 1407)))            # the whole window
    cases.append((rand(n, 1000, 1046), rand(n, 1000, 1046)))          # the magnitudes of a solve
    a, d = rand(n, 600, 1450), rand(n, 600, 1450)                     # across the edges of the window
    d[: n // 8] = (d[: n // 8].view(np.uint64) | np.uint64((1 << 52) - 1)).view(np.float64)            # mantissa all ones
    d[n // 8: n // 4] = (d[n // 8: n // 4].view(np.uint64) & ~np.uint64((1 << 52) - 1)).view(np.float64)  # powers of two
    a[n // 4: n // 2] = d[n // 4: n // 2] * rng.choice([1.0, -1.0, 3.0, 1.0 / 3.0], n // 4)           # equal / simple ratios
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                        1.0, -1.0, 2.0 ** -383, 2.0 ** 384, np.nextafter(2.0 ** 385, 0), 2.0 ** 385, np.nextafter(2.0 ** -383, 0)])
    k = len(special)
    a[-k * k:] = np.repeat(special, k); d[-k * k:] = np.tile(special, k)
    cases.append((a, d))
    inwin = 0
    for a, d in cases:
        fast, plain, win = probe(a, d)
        assert np.array_equal(fast.view(np.uint64), plain.view(np.uint64)), np.flatnonzero(fast.view(np.uint64) != plain.view(np.uint64))[:5]
        ok = np.isfinite(a) & np.isfinite(d) & (d != 0)
        with np.errstate(all="ignore"):
            host = a[ok] / d[ok]   # (IEEE division of the host: correctly rounded)
        assert np.array_equal(host.view(np.uint64), plain[ok].view(np.uint64))
        ea = ((a.view(np.uint64) >> np.uint64(52)) & np.uint64(0x7ff)).astype(np.int64)
        ed = ((d.view(np.uint64) >> np.uint64(52)) & np.uint64(0x7ff)).astype(np.int64)
        w = win.astype(bool)
        # where the short sequence's quotient was taken, both operands were inside the window in which it IS the division
        # (dividend [640, 1407], divisor [923, 1123]: trsv_syncfree.hip) ...
        assert np.all((ea[w] >= 640) & (ea[w] <= 1407) & (ed[w] >= 923) & (ed[w] <= 1123))
        # ... and it is taken wherever the operands are comfortably inside (not merely never wrong)
        sure = (ed >= 923) & (ed <= 1123) & (ea - ed >= -248) & (ea - ed <= 248)
        assert np.all(w[sure])
        inwin += int(win.sum())
    assert inwin > 4_000_000
    # the diagonals of the ILU(0) factors of the class, every numbering
    from oracle import oracle
    for kind in ("lex", "rcm", "morton", "random", "delaunay"):
        rp, ci, va = gen.shell_variant(60, kind)
        lu = oracle.ilu0(rp, ci, va)
        dg = lu[np.flatnonzero(ci == np.repeat(np.arange(len(rp) - 1), np.diff(rp)))]
        assert len(dg) == len(rp) - 1
        reps = 40
        d = np.tile(dg, reps)
        a = rng.standard_normal(len(d)) * 10.0 ** rng.integers(-8, 9, len(d))
        fast, plain, win = probe(a, d)
        assert win.all() and np.array_equal(fast.view(np.uint64), plain.view(np.uint64)), kind
