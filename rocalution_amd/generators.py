"""Synthetic operators used by the tests and the benchmark (host side, NumPy).

* ``poisson7(N)``  -- the 3-D 7-point Poisson matrix of BASELINE.json's configs. The reference
  has no 7-point generator (its 3-D generators are 27-point, clients/include/utility.hpp:110-177),
  so the layout is this repo's own: row ``r = (z*N + y)*N + x``, entries in
  ascending column order ``r-N², r-N, r-1, r (=6), r+1, r+N, r+N²`` where the neighbour exists,
  off-diagonals -1.  nnz = 7N³ - 6N².
* ``laplace2d(ndim)`` -- the 2-D 5-point Laplacian the reference's own solver tests use
  (behaviour of clients/include/utility.hpp:45-106 ``gen_2d_laplacian``: diag 4, neighbours -1).
* ``gr_30_30()`` -- exact re-synthesis of the Harwell-Boeing ``gr_30_30`` matrix of config 1:
  9-point 30x30 Laplacian, diag 8, the 8 surrounding neighbours -1 (900 rows, 7744 nnz after the
  MatrixMarket reader's symmetric expansion, src/base/host/host_io.cpp:218-272).

All generators return ``(row_offset[int32], col[int32], val[dtype])`` with sorted rows.
The device-side generator used by bench.py for 512³ lives in csrc (ramd_gen_poisson7).
"""
import numpy as np


def poisson7(N, dtype=np.float64):
    n = N * N * N
    r = np.arange(n, dtype=np.int64)
    x = r % N
    y = (r // N) % N
    z = r // (N * N)
    offs = np.array([-N * N, -N, -1, 0, 1, N, N * N], dtype=np.int64)
    mask = np.stack([z > 0, y > 0, x > 0, np.ones(n, bool), x < N - 1, y < N - 1, z < N - 1], axis=1)
    cols = r[:, None] + offs[None, :]
    vals = np.where(offs == 0, 6.0, -1.0).astype(dtype)
    vals = np.broadcast_to(vals[None, :], (n, 7))
    counts = mask.sum(axis=1)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rp[1:])
    assert rp[-1] < 2**31
    return rp.astype(np.int32), cols[mask].astype(np.int32), np.ascontiguousarray(vals[mask])


def laplace2d(ndim, dtype=np.float64):
    n = ndim * ndim
    r = np.arange(n, dtype=np.int64)
    i = r % ndim
    j = r // ndim
    offs = np.array([-ndim, -1, 0, 1, ndim], dtype=np.int64)
    mask = np.stack([j > 0, i > 0, np.ones(n, bool), i < ndim - 1, j < ndim - 1], axis=1)
    cols = r[:, None] + offs[None, :]
    vals = np.broadcast_to(np.where(offs == 0, 4.0, -1.0).astype(dtype)[None, :], (n, 5))
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(mask.sum(axis=1), out=rp[1:])
    return rp.astype(np.int32), cols[mask].astype(np.int32), np.ascontiguousarray(vals[mask])


def gr_30_30(dtype=np.float64):
    m = 30
    n = m * m
    rows = []
    for j in range(m):
        for i in range(m):
            ent = []
            for dj in (-1, 0, 1):
                for di in (-1, 0, 1):
                    ii, jj = i + di, j + dj
                    if 0 <= ii < m and 0 <= jj < m:
                        ent.append((jj * m + ii, 8.0 if (di == 0 and dj == 0) else -1.0))
            ent.sort()
            rows.append(ent)
    rp = np.zeros(n + 1, dtype=np.int32)
    ci, va = [], []
    for r, ent in enumerate(rows):
        for c, v in ent:
            ci.append(c)
            va.append(v)
        rp[r + 1] = len(ci)
    return rp, np.array(ci, dtype=np.int32), np.array(va, dtype=dtype)


def random_sparse(n, avg_nnz, seed, dtype=np.float64, diag_dominant=True, max_row=None):
    """Irregular test matrix: ragged rows (some empty off-diagonals, some long), sorted columns,
    always a stored diagonal; diagonally dominant so ILU(0)/Jacobi are well defined."""
    rng = np.random.default_rng(seed)
    rp = [0]
    ci, va = [], []
    for r in range(n):
        k = int(rng.poisson(avg_nnz))
        if max_row is not None:
            k = min(k, max_row - 1)
        if r % 97 == 0:
            k = 0
        if r % 211 == 5 and max_row is None:
            k = min(n - 1, 12 * avg_nnz)
        cols = set(rng.integers(0, n, size=k).tolist()) if k > 0 else set()
        cols.discard(r)
        cols = sorted(cols | {r})
        vals = rng.uniform(-1.0, 1.0, size=len(cols))
        if diag_dominant:
            s = np.abs(vals).sum()
            vals[cols.index(r)] = s + 1.0
        ci.extend(cols)
        va.extend(vals.tolist())
        rp.append(len(ci))
    return np.array(rp, dtype=np.int32), np.array(ci, dtype=np.int32), np.array(va, dtype=dtype)


def to_scipy(rp, ci, va):
    import scipy.sparse as sp
    n = len(rp) - 1
    return sp.csr_matrix((va, ci, rp), shape=(n, n))
