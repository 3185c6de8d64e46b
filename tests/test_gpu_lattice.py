"""The lattice form of the sparse triangular solve (trsv_lattice.hip: pencils marched along x, one wave per 8 x 8 cross-section).

Contract: the same operations per row in the same order as the host loops (host_matrix_csr.cpp:1163-1221 LUSolve,
:1357-1404 LSolve, :1420-1466 USolve) => bit-identical results with the level-scheduled kernel, the box tiles, the oracle and
the goldens.  Covered here:
  * grids whose extents are / are not multiples of the 8 x 8 cross-section and of the 16-element staging blocks, odd nx (the
    8-byte staging path), fewer than 8 planes, unsymmetric values; fp64 and fp32; LUSolve, LSolve / USolve with and without
    the unit diagonal; against the level-scheduled kernel (RAMD_TRSV_LAT=0 + RAMD_TRSV_CT=0) and, where it is fast, the oracle
  * 77^3 and 100^3 ("lusolve goldens + the forced-box-tile suites + 77^3 / 100^3 bit-exact")
  * the whole triangular / preconditioner / solver-history parity selection of the GPU suite once more with the form forced
    onto every lattice it recognises, however small (RAMD_TRSV_LAT=2: the `poisson8` goldens run through it)
  * 512^3: lattice form == box tiles bit for bit, ten solves in a row (hand-offs between 4096 pencils under load)
  * what the form refuses (9-point stencil, a missing entry, a permuted numbering) falls through to the general plans
  * the plan statistics hook
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lattice_csr(nx, ny, nz, seed=0):
    """7-point pattern on nx x ny x nz (x fastest), unsymmetric values, diagonally dominant"""
    rng = np.random.default_rng(seed)

    def lap(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    A = sp.kron(sp.eye(nz), sp.kron(sp.eye(ny), lap(nx))) + sp.kron(sp.eye(nz), sp.kron(lap(ny), sp.eye(nx))) \
        + sp.kron(lap(nz), sp.kron(sp.eye(ny), sp.eye(nx)))
    A = A.tocsr(); A.sort_indices()
    A.data = A.data * rng.uniform(0.5, 1.5, A.nnz)
    A = (A + sp.diags(np.full(A.shape[0], 3.0))).tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


_SOLVES = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import rocalution_amd as ra
from rocalution_amd import capi
from test_gpu_lattice import lattice_csr
ra.init_rocalution()
lib = capi.load()
import ctypes as C
out = {}
want_form = int(sys.argv[2])
for (nx, ny, nz) in %(grids)s:
    rp, ci, va = lattice_csr(nx, ny, nz)
    n = len(rp) - 1
    for dt in (np.float64, np.float32):
        tag = "%%dx%%dx%%d_%%s" %% (nx, ny, nz, np.dtype(dt).name)
        b = np.random.default_rng(7).uniform(-1, 1, n).astype(dt)
        A = ra.LocalMatrix(dt); A.SetDataPtrCSR(rp, ci, va.astype(dt))
        A.ILU0Factorize(); A.LUAnalyse()
        st = (C.c_int64 * 16)()
        capi.check(lib.ramd_tri_plan_stats(0, st)); form_l = st[0]
        capi.check(lib.ramd_tri_plan_stats(1, st)); form_u = st[0]
        assert (form_l == 4) == (want_form == 4) and form_l == form_u, (tag, form_l, form_u)
        if want_form == 4:
            assert (st[9], st[10], st[11]) == (nx, ny, nz), list(st)
        y = ra.LocalVector(dt); y.Allocate("", n)
        for rep in range(3):  # (the faces of a plan are reused from solve to solve)
            A.LUSolve(ra.LocalVector(dt, data=b), y)
        out[tag + "_lu"] = y.numpy().copy()
        out[tag + "_factors"] = A.CopyToCSR()[2]
        B = ra.LocalMatrix(dt); B.SetDataPtrCSR(rp, ci, va.astype(dt))
        for unit in (False, True):
            B.LAnalyse(unit); B.LSolve(ra.LocalVector(dt, data=b), y); out[tag + "_l%%d" %% unit] = y.numpy().copy()
            B.UAnalyse(unit); B.USolve(ra.LocalVector(dt, data=b), y); out[tag + "_u%%d" %% unit] = y.numpy().copy()
np.savez(sys.argv[1], **out)
"""

GRIDS_SMALL = [(8, 8, 8), (16, 8, 8), (20, 20, 20), (32, 9, 17), (77, 13, 10), (33, 24, 16), (64, 64, 16), (100, 30, 30),
               (16, 40, 3), (9, 5, 2), (130, 8, 9)]


def _run_solves(tmp_path, name, grids, env, form):
    script = tmp_path / "lat_solves.py"
    script.write_text(_SOLVES % {"root": ROOT, "grids": repr(grids)})
    out = str(tmp_path / (name + ".npz"))
    p = subprocess.run([sys.executable, str(script), out, str(form)], cwd=ROOT, env=dict(os.environ, **env), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]
    return dict(np.load(out)), p.stdout


def _same_bits(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_lattice_form_equals_the_level_scheduled_kernel_bit_for_bit(tmp_path, oracle):
    level, _ = _run_solves(tmp_path, "level", GRIDS_SMALL, {"RAMD_TRSV_LAT": "0", "RAMD_TRSV_CT": "0"}, 1)
    lat, log = _run_solves(tmp_path, "lat", GRIDS_SMALL, {"RAMD_TRSV_LAT": "2", "RAMD_TRSV_CT_VERBOSE": "1"}, 4)
    assert "lattice plan (lower)" in log and "lattice plan (upper)" in log
    assert sorted(level) == sorted(lat) and len(lat) == len(GRIDS_SMALL) * 2 * 6
    for k in level:
        # (bit patterns: a unit-diagonal solve of these operators overflows in fp32 on the larger grids -- the same
        #  infinities and NaNs have to come out of both kernels)
        assert _same_bits(level[k], lat[k]), k
    for (nx, ny, nz) in GRIDS_SMALL:
        if nx * ny * nz > 40000:
            continue
        rp, ci, va = lattice_csr(nx, ny, nz)
        b = np.random.default_rng(7).uniform(-1, 1, len(rp) - 1)
        tag = "%dx%dx%d_float64" % (nx, ny, nz)
        lu = oracle.ilu0(rp, ci, va)
        assert np.array_equal(lu, lat[tag + "_factors"]), tag
        assert np.array_equal(oracle.lusolve(rp, ci, lu, b), lat[tag + "_lu"]), tag
        assert np.array_equal(oracle.lsolve(rp, ci, va, b, False), lat[tag + "_l0"]), tag
        assert np.array_equal(oracle.usolve(rp, ci, va, b, False), lat[tag + "_u0"]), tag


def test_77_and_100_cubed_bit_exact(tmp_path):
    grids = [(77, 77, 77), (100, 100, 100)]
    level, _ = _run_solves(tmp_path, "level", grids, {"RAMD_TRSV_LAT": "0", "RAMD_TRSV_CT": "0"}, 1)
    tiles, _ = _run_solves(tmp_path, "tiles", grids, {"RAMD_TRSV_LAT": "0"}, 2)
    lat, _ = _run_solves(tmp_path, "lat", grids, {}, 4)  # (the default takes the lattice form at these sizes)
    for k in level:
        assert _same_bits(level[k], lat[k]) and _same_bits(tiles[k], lat[k]), k


SELECT = ("(ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0) "
          "and not full_size and not cpp and not fresh_process")


def forced_job():
    env = dict(os.environ, RAMD_TRSV_LAT="2", RAMD_TRSV_CT_VERBOSE="1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_kernels.py"), os.path.join(ROOT, "tests", "test_gpu_solvers.py"), "-k", SELECT]
    return cmd, env, 1500


def test_parity_suite_with_the_lattice_form_forced():
    """every lattice the suite's matrices contain (the `poisson8` goldens, the Poisson systems of the oracle comparisons) goes
    through the pencil kernel, whatever its size; everything else keeps its plan"""
    from conftest import forced_run
    from test_gpu_box_tiles_forced import _tri_family
    rc, out = forced_run("tri", "lattice", _tri_family())
    tail = out[-3000:]
    assert rc == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    assert "lattice plan (lower): 8 x 8 x 8" in out and "lattice plan (upper): 8 x 8 x 8" in out, tail


def test_what_is_not_a_lattice_keeps_the_general_plans():
    import rocalution_amd as ra
    from rocalution_amd import capi, generators as gen
    ra.init_rocalution()
    lib = capi.load()
    os.environ["RAMD_TRSV_LAT"] = "2"
    try:
        st = (C.c_int64 * 16)()

        def form(rp, ci, va):
            A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
            A.LUAnalyse()
            capi.check(lib.ramd_tri_plan_stats(0, st))
            return st[0]
        rp, ci, va = lattice_csr(12, 10, 9)
        assert form(rp, ci, va) == 4
        # one entry removed
        A = sp.csr_matrix((va, ci, rp)).tolil()
        A[500, 499] = 0.0
        A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
        assert form(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data) != 4
        # an extra entry
        A = sp.csr_matrix((va, ci, rp)).tolil()
        A[500, 3] = 0.25
        A = A.tocsr(); A.sort_indices()
        assert form(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data) != 4
        # the same operator in a random numbering
        n = len(rp) - 1
        perm = np.random.default_rng(3).permutation(n)
        A = sp.csr_matrix((va, ci, rp))[perm][:, perm].tocsr(); A.sort_indices()
        assert form(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data) != 4
        # 9-point stencil in 2-D (gr_30_30)
        assert form(*gen.gr_30_30()) != 4
        # periodic wrap in x: r - 1 present at x = 0 would be a wrong neighbour
        A = sp.csr_matrix((va, ci, rp)).tolil()
        A[12, 11] = -0.5  # (x = 0 of the second line linked to x = 11 of the first)
        A = A.tocsr(); A.sort_indices()
        assert form(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data) != 4
    finally:
        del os.environ["RAMD_TRSV_LAT"]


def test_512_cubed_lattice_equals_box_tiles_and_soak(tmp_path):
    script = tmp_path / "soak.py"
    script.write_text(r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import rocalution_amd as ra
ra.init_rocalution()
N = 512; n = N ** 3
A = ra.LocalMatrix(); A.GenPoisson7(N)
A.ILU0Factorize()
b = ra.LocalVector(data=np.random.default_rng(1).uniform(-1.0, 1.0, n))
x = ra.LocalVector(); x.Allocate("", n)
res = []
for mode in ("0", "1"):
    os.environ["RAMD_TRSV_LAT"] = mode
    A.LUAnalyse()
    for rep in range(10 if mode == "1" else 1):
        x.Zeros()
        A.LUSolve(b, x)
        res.append(x.numpy().copy())
    A.LUAnalyseClear()
for r in res[1:]:
    assert np.array_equal(res[0].view(np.uint8), r.view(np.uint8))
assert np.isfinite(res[0]).all()
print("soak ok", len(res))
""" % ROOT)
    p = subprocess.run([sys.executable, str(script)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=1200)
    assert p.returncode == 0 and "soak ok 11" in p.stdout, p.stdout[-3000:]


# ------------------------------------------------------------------ the one-pass red-black form of the MC-SGS apply (k_mc_rb)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_red_black_lattice_sgs_apply_equals_the_block_form(dtype):
    """mcsgs.hip k_mc_rb: on a 7- / 5-point lattice operator whose two colours are the parities of x + y + z, ONE pass over both
    colours (32 x 32 tiles marched along z, three planes of each kind in LDS, the one-cell ring of colour-1 values recomputed)
    replaces the colour sweeps.  Same operations per row in the same order as the reference's block form
    (preconditioner_multicolored_gs.cpp:127-215): bit-identical, on grids that are / are not multiples of the tile, odd extents,
    one plane, fewer planes than a chunk, unsymmetric values."""
    import rocalution_amd as ra
    from rocalution_amd import solvers as S
    ra.init_rocalution()
    os.environ["RAMD_MC_RB"] = "2"
    try:
        for (nx, ny, nz) in ((8, 8, 8), (33, 31, 5), (64, 40, 70), (100, 7, 3), (17, 65, 66), (40, 36, 1), (6, 5, 130)):
            rp, ci, va = lattice_csr(nx, ny, nz, seed=nx + ny)
            n = len(rp) - 1
            A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va.astype(dtype))
            x = ra.LocalVector(dtype, data=np.random.default_rng(3).uniform(-1, 1, n).astype(dtype))
            res = []
            for fused in (True, False):
                pc = S.MultiColoredSGS()
                pc.SetFusedSweeps(fused)
                ls = S.BiCGStab(dtype); ls.SetOperator(A); ls.SetPreconditioner(pc); ls.Build()
                assert ls.GetNumColors() == 2
                z = ra.LocalVector(dtype); z.Allocate("", n)
                for rep in range(2):
                    ls.PrecondApply(x, z)
                res.append(z.numpy().copy())
                ls.Clear()
            assert np.isfinite(res[0]).all() and _same_bits(res[0], res[1]), (nx, ny, nz, np.dtype(dtype).name)
    finally:
        del os.environ["RAMD_MC_RB"]


RB_GRIDS = ((8, 8, 8), (33, 31, 5), (64, 40, 70), (100, 7, 3), (17, 65, 66), (40, 36, 1), (6, 5, 130))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_red_black_lattice_sgs_apply_equals_the_oracle(dtype, oracle):
    """k_mc_rb compared DIRECTLY with the CPU restatement of the reference's MultiColoredSGS apply
    (preconditioner_multicolored_gs.cpp:127-215 on the permuted blocks of preconditioner_multicolored.cpp:303-413), not through
    the colour sweeps of this library: fp64 bit for bit against the fp64 oracle, fp32 against the fp32 oracle; the plan's own
    answer (ramd_mcsgs_info) says that the red-black form is what ran, whatever the size of the grid."""
    import rocalution_amd as ra
    from rocalution_amd import capi, solvers as S
    ra.init_rocalution()
    lib = capi.load()
    os.environ["RAMD_MC_RB"] = "2"
    try:
        for (nx, ny, nz) in RB_GRIDS:
            rp, ci, va = lattice_csr(nx, ny, nz, seed=nx + ny)
            va = va.astype(dtype)
            n = len(rp) - 1
            rhs = np.random.default_rng(3).uniform(-1, 1, n).astype(dtype)
            A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va)
            ls = S.BiCGStab(dtype); ls.SetOperator(A); ls.SetPreconditioner(S.MultiColoredSGS()); ls.Build()
            st = (C.c_int64 * 8)()
            capi.check(lib.ramd_mcsgs_info(None, st))
            assert st[0] == 2 and (st[4], st[5], st[6]) == (nx, ny, nz) and st[7] == n, list(st)
            z = ra.LocalVector(dtype); z.Allocate("", n)
            for rep in range(2):
                ls.PrecondApply(ra.LocalVector(dtype, data=rhs), z)
            want = oracle.precond_apply(oracle.PC_MCSGS, rp, ci, va, rhs)
            assert want.dtype == np.dtype(dtype) and _same_bits(want, z.numpy()), (nx, ny, nz, np.dtype(dtype).name)
            ls.Clear()
    finally:
        del os.environ["RAMD_MC_RB"]


def test_red_black_form_is_refused_where_the_sweeps_would_subtract_in_another_order(oracle):
    """a colouring permutation that is valid but does not keep the natural order inside a colour makes the colour
    sweeps subtract in another order than k_mc_rb's ascending natural offsets; the plan must fall back to the sweeps then.  The
    lattice operator in a numbering with two lines swapped is such a case: still two colours, no longer an x-fastest lattice."""
    import rocalution_amd as ra
    from rocalution_amd import capi, solvers as S
    ra.init_rocalution()
    lib = capi.load()
    os.environ["RAMD_MC_RB"] = "2"
    try:
        nx, ny, nz = 12, 10, 6
        rp, ci, va = lattice_csr(nx, ny, nz, seed=5)
        n = len(rp) - 1
        perm = np.arange(n)
        perm[[nx * 3 + 2, nx * 3 + 4]] = perm[[nx * 3 + 4, nx * 3 + 2]]  # (two cells of one colour change places)
        M = sp.csr_matrix((va, ci, rp))[perm][:, perm].tocsr(); M.sort_indices()
        rp2, ci2, va2 = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data
        rhs = np.random.default_rng(4).uniform(-1, 1, n)
        A = ra.LocalMatrix(); A.SetDataPtrCSR(rp2, ci2, va2)
        ls = S.BiCGStab(); ls.SetOperator(A); ls.SetPreconditioner(S.MultiColoredSGS()); ls.Build()
        st = (C.c_int64 * 8)()
        capi.check(lib.ramd_mcsgs_info(None, st))
        assert st[0] != 2, list(st)
        z = ra.LocalVector(); z.Allocate("", n)
        ls.PrecondApply(ra.LocalVector(data=rhs), z)
        assert _same_bits(oracle.precond_apply(oracle.PC_MCSGS, rp2, ci2, va2, rhs), z.numpy())
    finally:
        del os.environ["RAMD_MC_RB"]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_interleaved_analyses_on_a_lattice_keep_their_own_plans(dtype, oracle):
    """LUAnalyse and LAnalyse / UAnalyse are independent analyses with their own diagonal flags; a lattice
    plan fixes the flag when it is built.  Every legal interleaving on a lattice of >= 4096 rows (the size from which the pencil
    form is taken by default) must give what the same solve gives alone."""
    import rocalution_amd as ra
    ra.init_rocalution()
    nx, ny, nz = 20, 18, 13  # 4680 rows
    rp, ci, va = lattice_csr(nx, ny, nz, seed=2)
    va = va.astype(dtype)
    n = len(rp) - 1
    b = np.random.default_rng(9).uniform(-1, 1, n).astype(dtype)
    bv = ra.LocalVector(dtype, data=b)
    lu = oracle.ilu0(rp, ci, va)
    want = dict(lu=oracle.lusolve(rp, ci, lu, b))
    for unit in (False, True):
        want["l%d" % unit] = oracle.lsolve(rp, ci, lu, b, unit)
        want["u%d" % unit] = oracle.usolve(rp, ci, lu, b, unit)
    A = ra.LocalMatrix(dtype); A.SetDataPtrCSR(rp, ci, va)
    A.ILU0Factorize()
    y = ra.LocalVector(dtype); y.Allocate("", n)

    def check(what):
        {"lu": A.LUSolve, "l": A.LSolve, "u": A.USolve}[what.rstrip("01")](bv, y)
        assert _same_bits(want[what], y.numpy()), what

    A.LAnalyse(False); A.LUAnalyse(); check("l0"); check("lu")  # (the ADVICE sequence: LSolve skipped the division)
    A.UAnalyse(True); check("u1"); check("lu"); check("l0")
    A.LUAnalyseClear()
    A.LUAnalyse(); A.LAnalyse(True); A.UAnalyse(False); check("lu"); check("l1"); check("u0")
    A.LAnalyseClear(); check("lu"); check("u0")  # (LAnalyseClear after LUAnalyse left a half plan behind)
    A.UAnalyseClear(); check("lu")
    A.LAnalyse(False); A.UAnalyse(True); check("lu"); check("l0"); check("u1")
    os.environ["RAMD_TRSV_LAT"] = "0"  # (the form changes between two analyses of one matrix)
    try:
        A.LAnalyse(True); check("l1"); check("lu"); check("u1")
        A.LUAnalyse(); check("lu"); check("l1"); check("u1")
    finally:
        del os.environ["RAMD_TRSV_LAT"]
    A.LUAnalyseClear()
