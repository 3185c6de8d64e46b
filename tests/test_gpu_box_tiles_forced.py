"""The record-form box-tile triangular solve (k_trsv_rec, trisolve.hip) on the SMALL parity matrices.

By default a plan takes the box-tile form only for matrices of at least 4096 rows with chains of mean length >= 8, and
external values are de-duplicated only where references outnumber rows -- the goldens (30x30 grids, random patterns) and
most oracle comparisons would exercise only the level-scheduled fallback.  Here the same bit-exact tests run again in a
fresh process with the box-tile form forced for every matrix (RAMD_TRSV_CT_MINROWS / _MINLEN = 0), once with every external
value listed per reference and once de-duplicated (RAMD_TRSV_CT_DEDUP = 0 / 1): ILU(0)/IC factors + LUSolve / LLSolve /
LSolve / USolve vs the goldens (host_matrix_csr.cpp:1163-1221, :1344-1466), LUSolve on Poisson 24^3 / 40^3 vs the oracle,
preconditioner applications, the solver histories vs the goldens, and the af_shell10-class surrogate (8 lanes per row):
~300 tests per run.  The settings are read once per process,
hence the subprocess."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SELECT = ("(ilu or lusolve or trisolve or preconditioner_apply or sgs or solvers_vs_golden or rebuild_numeric or gmres30_ilu0) "
          "and not full_size and not cpp and not fresh_process")


def _tri_family():
    """the five forced-form passes over the triangular-solve parity selection (box tiles x 3, lattice pencils, sync-free groups):
    started together, see conftest.forced_run"""
    jobs = {}
    for dedup in ("0", "1", "0w"):
        # "0w": additionally the opt-in layout of w (a tile's values placed by consumer, RAMD_TRSV_WSLOT=1) and 8-byte index pairs
        env = dict(os.environ, RAMD_TRSV_CT_MINROWS="0", RAMD_TRSV_CT_MINLEN="0", RAMD_TRSV_CT_DEDUP=dedup[0],
                   RAMD_TRSV_CT_VERBOSE="1")
        if dedup.endswith("w"):
            env.update(RAMD_TRSV_WSLOT="1", RAMD_TRSV_PACK="0", RAMD_TRSV_CLASSES="0")
        cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider",
               os.path.join(ROOT, "tests", "test_gpu_kernels.py"), os.path.join(ROOT, "tests", "test_gpu_solvers.py"),
               os.path.join(ROOT, "tests", "test_gpu_shell.py"), "-k", SELECT]
        jobs["tiles" + dedup] = (cmd, env, 1500)
    import test_gpu_lattice as TL
    import test_gpu_syncfree as TS
    jobs["lattice"] = TL.forced_job()
    jobs["syncfree"] = TS.forced_job()
    return jobs


@pytest.mark.gpu
@pytest.mark.parametrize("dedup", ["0", "1", "0w"])
def test_parity_suite_with_box_tiles_forced(dedup):
    from conftest import forced_run
    rc, out = forced_run("tri", "tiles" + dedup, _tri_family())
    tail = out[-3000:]
    assert rc == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
    # the forced form was really used (the plan builder reports every box-tile plan it makes)
    assert "box-tile plan (lower)" in out and "box-tile plan (upper)" in out, tail
    if dedup == "1":
        assert "distinct values" in out, tail


_CROSS = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import rocalution_amd as ra
from rocalution_amd import generators as gen
ra.init_rocalution()
out = {}
for tag, (rp, ci, va) in (("poisson20", gen.poisson7(20)), ("shell16", gen.shell_surrogate(16, 16))):
    n = len(rp) - 1
    for dt in (np.float64, np.float32):
        A = ra.LocalMatrix(dt); A.SetDataPtrCSR(rp, ci, va.astype(dt))
        A.ILU0Factorize(); A.LUAnalyse()
        b = np.random.default_rng(7).uniform(-1, 1, n).astype(dt)
        y = ra.LocalVector(dt); y.Allocate("", n)
        for rep in range(3):
            A.LUSolve(ra.LocalVector(dt, data=b), y)
        out["%%s_%%s_lu" %% (tag, np.dtype(dt).name)] = y.numpy().copy()
        B = ra.LocalMatrix(dt); B.SetDataPtrCSR(rp, ci, va.astype(dt))
        B.LAnalyse(False); B.LSolve(ra.LocalVector(dt, data=b), y)
        out["%%s_%%s_l" %% (tag, np.dtype(dt).name)] = y.numpy().copy()
        B.UAnalyse(False); B.USolve(ra.LocalVector(dt, data=b), y)
        out["%%s_%%s_u" %% (tag, np.dtype(dt).name)] = y.numpy().copy()
np.savez(sys.argv[1], **out)
"""


@pytest.mark.gpu
def test_fp32_and_fp64_box_tiles_equal_the_level_scheduled_kernel(tmp_path):
    """Both kernels perform the host loop's operations per row in the same order (host_matrix_csr.cpp:1163-1221), so their
    results are bit-identical -- in fp32 as in fp64, one lane per row (7-point operator) and eight (shell surrogate).  The
    fp32 instantiations have no golden of their own; this pins them to the form the goldens pin."""
    import numpy as np
    script = tmp_path / "cross.py"
    script.write_text(_CROSS % {"root": ROOT})
    res = {}
    for tag, env in (("level", {"RAMD_TRSV_CT": "0"}),
                     ("tiles", {"RAMD_TRSV_CT_MINROWS": "0", "RAMD_TRSV_CT_MINLEN": "0", "RAMD_TRSV_CT_DEDUP": "1",
                                "RAMD_TRSV_CT_VERBOSE": "1"})):
        out = str(tmp_path / (tag + ".npz"))
        p = subprocess.run([sys.executable, str(script), out], cwd=ROOT, env=dict(os.environ, **env), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-3000:]
        if tag == "tiles":
            assert "box-tile plan (lower)" in p.stdout, p.stdout[-3000:]
        else:
            assert "box-tile plan" not in p.stdout
        res[tag] = dict(np.load(out))
    assert sorted(res["level"]) == sorted(res["tiles"]) and len(res["level"]) == 12
    for k in res["level"]:
        a, b = res["level"][k], res["tiles"][k]
        assert a.dtype == b.dtype and np.isfinite(a).all(), k
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


_RAGGED = r"""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, %(root)r)
import rocalution_amd as ra
from oracle import oracle
ra.init_rocalution()
rng = np.random.default_rng(11)


def banded(n, wmax, reach):
    # symmetric pattern, row i linked to i-1 (chains) and to a random number of rows within `reach`; diagonally dominant
    rows, cols = [], []
    for i in range(1, n):
        k = int(rng.integers(0, wmax))
        js = set([i - 1]) | set(int(j) for j in rng.integers(max(0, i - reach), i, size=k))
        for j in js:
            rows += [i, j]; cols += [j, i]
    v = rng.uniform(-1, -0.05, len(rows))
    A = sp.coo_matrix((v, (rows, cols)), shape=(n, n)).tocsr(); A.sum_duplicates()
    A = (A + A.T) * 0.5
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
    A = A.tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


checked = 0
for n, wmax, reach in ((1, 1, 1), (2, 1, 1), (65, 2, 8), (777, 3, 40), (3000, 8, 100), (3000, 20, 64), (2500, 31, 200),
                       (1500, 60, 300)):
    rp, ci, va = banded(n, wmax, reach)
    low = max(int(((ci[rp[i]:rp[i + 1]]) < i).sum()) for i in range(n))
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    A.ILU0Factorize()
    lu = oracle.ilu0(rp, ci, va)
    assert np.array_equal(A.CopyToCSR()[2], lu), ("ilu0", n, wmax)
    A.LUAnalyse()
    y = ra.LocalVector(); y.Allocate("", n)
    for rep in range(2):
        b = rng.uniform(-1, 1, n)
        A.LUSolve(ra.LocalVector(data=b), y)
        assert np.array_equal(y.numpy(), oracle.lusolve(rp, ci, lu, b)), ("lusolve", n, wmax, low)
    B = ra.LocalMatrix(); B.SetDataPtrCSR(rp, ci, va)
    B.LAnalyse(False); B.LSolve(ra.LocalVector(data=b), y)
    assert np.array_equal(y.numpy(), oracle.lsolve(rp, ci, va, b, False)), ("lsolve", n, wmax, low)
    B.UAnalyse(False); B.USolve(ra.LocalVector(data=b), y)
    assert np.array_equal(y.numpy(), oracle.usolve(rp, ci, va, b, False)), ("usolve", n, wmax, low)
    print("ok n=%%d longest lower row %%d" %% (n, low), flush=True)
    checked += 1
print("checked", checked)
"""


@pytest.mark.gpu
@pytest.mark.parametrize("dedup", ["0", "1"])
def test_ragged_rows_from_1_to_60_entries_bit_exact(tmp_path, dedup):
    """Random banded systems with rows of 1 .. 60 strictly-triangular entries: one lane per row (<= 8 entries), eight lanes
    per row (<= 32), and the level-scheduled fallback beyond; n = 1, 2, a partial wave, several tiles.  ILU(0) factors,
    LUSolve, LSolve and USolve against the oracle (host_matrix_csr.cpp:1163-1221, :1344-1466, :2096-2171), bit for bit."""
    script = tmp_path / "ragged.py"
    script.write_text(_RAGGED % {"root": ROOT})
    env = dict(os.environ, RAMD_TRSV_CT_MINROWS="0", RAMD_TRSV_CT_MINLEN="0", RAMD_TRSV_CT_DEDUP=dedup,
               RAMD_TRSV_CT_VERBOSE="1")
    p = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-4000:]
    assert "checked 8" in p.stdout, p.stdout[-4000:]
    assert "wmax=" in p.stdout and "box-tile plan (lower)" in p.stdout, p.stdout[-2000:]
