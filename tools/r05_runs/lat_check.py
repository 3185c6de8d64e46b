"""Lattice (pencil) triangular solve against the level-scheduled kernel and the oracle, several grids; then timings."""
import os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocalution_amd as ra
from oracle import oracle
ra.init_rocalution()


def lattice(nx, ny, nz, seed=0):
    rng = np.random.default_rng(seed)
    def lap(n):
        return sp.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])
    A = sp.kron(sp.eye(nz), sp.kron(sp.eye(ny), lap(nx))) + sp.kron(sp.eye(nz), sp.kron(lap(ny), sp.eye(nx))) \
        + sp.kron(lap(nz), sp.kron(sp.eye(ny), sp.eye(nx)))
    A = A.tocsr(); A.sort_indices()
    A.data = A.data * rng.uniform(0.5, 1.5, A.nnz)  # unsymmetric values, same pattern
    A = A + sp.diags(np.full(A.shape[0], 3.0))
    A = A.tocsr(); A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def run(mode, rp, ci, va, dt, b):
    os.environ["RAMD_TRSV_LAT"] = mode
    if mode == "0":
        os.environ["RAMD_TRSV_CT"] = "0"
    n = len(rp) - 1
    res = {}
    A = ra.LocalMatrix(dt); A.SetDataPtrCSR(rp, ci, va.astype(dt))
    A.ILU0Factorize(); A.LUAnalyse()
    y = ra.LocalVector(dt); y.Allocate("", n)
    for rep in range(3):
        A.LUSolve(ra.LocalVector(dt, data=b), y)
    res["lu"] = y.numpy().copy()
    B = ra.LocalMatrix(dt); B.SetDataPtrCSR(rp, ci, va.astype(dt))
    for unit in (False, True):
        B.LAnalyse(unit); B.LSolve(ra.LocalVector(dt, data=b), y); res["l%d" % unit] = y.numpy().copy()
        B.UAnalyse(unit); B.USolve(ra.LocalVector(dt, data=b), y); res["u%d" % unit] = y.numpy().copy()
    return res


bad = 0
for (nx, ny, nz) in ((8, 8, 8), (16, 8, 8), (20, 20, 20), (32, 9, 17), (77, 13, 10), (33, 24, 16), (64, 64, 16), (100, 30, 30), (16, 40, 3)):
    rp, ci, va = lattice(nx, ny, nz)
    n = len(rp) - 1
    for dt in (np.float64, np.float32):
        b = np.random.default_rng(7).uniform(-1, 1, n).astype(dt)
        r0 = run("0", rp, ci, va, dt, b)
        r2 = run("2", rp, ci, va, dt, b)
        for k in r0:
            ok = np.array_equal(r0[k], r2[k])
            if not ok:
                bad += 1
                d = np.abs(r0[k].astype(np.float64) - r2[k].astype(np.float64))
                print("MISMATCH", (nx, ny, nz), np.dtype(dt).name, k, "max", d.max(), "first", int(np.argmax(d > 0)), "count", int((d > 0).sum()), flush=True)
        if dt == np.float64 and n <= 30000:
            lu = oracle.ilu0(rp, ci, va)
            ref = oracle.lusolve(rp, ci, lu, b)
            if not np.array_equal(ref, r2["lu"]):
                bad += 1
                print("ORACLE MISMATCH", (nx, ny, nz))
    print("grid", (nx, ny, nz), "done", flush=True)
print("mismatches", bad)
if len(sys.argv) > 1:
    N = int(sys.argv[1])
    from rocalution_amd import generators as gen
    for shape in ((N, N, N), (N, N, max(8, N // 8))):
        rp, ci, va = lattice(*shape)
        n = len(rp) - 1
        for mode in ("0c", "1"):
            os.environ["RAMD_TRSV_LAT"] = mode[0]
            os.environ.pop("RAMD_TRSV_CT", None)
            A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
            A.ILU0Factorize()
            t0 = time.time(); A.LUAnalyse(); ra.sync(); ta = time.time() - t0
            b = ra.LocalVector(data=np.ones(n)); y = ra.LocalVector(); y.Allocate("", n)
            for rep in range(3):
                A.LUSolve(b, y)
            ra.sync(); t0 = time.time()
            for rep in range(20):
                A.LUSolve(b, y)
            ra.sync(); t = (time.time() - t0) / 20
            print("shape", shape, "mode", mode, "analyse %.3f s, LUSolve %.3f ms" % (ta, t * 1e3), "checksum", float(y.numpy().sum()), flush=True)
            del A
