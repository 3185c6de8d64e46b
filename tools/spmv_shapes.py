"""SpMV throughput on operator shapes other than the 7-point stencil (GPU box):
27-point stencil (27 nnz/row, three far bands per plane), a banded FEM-like matrix with ~35 nnz/row, and a
random sparse matrix with scattered columns (worst case for the x gathers)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import rocalution_amd as ra  # noqa: E402
from rocalution_amd import capi  # noqa: E402

ra.init_rocalution()
lib = capi.load()


def stencil27(N):
    n = N ** 3
    r = np.arange(n, dtype=np.int64)
    x, y, z = r % N, (r // N) % N, r // (N * N)
    cols, masks = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                m = (x + dx >= 0) & (x + dx < N) & (y + dy >= 0) & (y + dy < N) & (z + dz >= 0) & (z + dz < N)
                cols.append(r + dz * N * N + dy * N + dx); masks.append(m)
    cols = np.stack(cols, axis=1); masks = np.stack(masks, axis=1)
    rp = np.zeros(n + 1, np.int64); np.cumsum(masks.sum(axis=1), out=rp[1:])
    ci = cols[masks].astype(np.int32)
    va = np.where(ci == np.repeat(r, np.diff(rp)).astype(np.int32), 26.0, -1.0)
    return rp.astype(np.int32), ci, va


def banded(n, per_row, half_band, seed):
    rng = np.random.default_rng(seed)
    off = np.sort(rng.choice(np.arange(-half_band, half_band + 1), size=per_row - 1, replace=False))
    off = np.unique(np.concatenate([off, [0]]))
    r = np.arange(n, dtype=np.int64)
    cols = r[:, None] + off[None, :]
    m = (cols >= 0) & (cols < n)
    rp = np.zeros(n + 1, np.int64); np.cumsum(m.sum(axis=1), out=rp[1:])
    ci = cols[m].astype(np.int32)
    va = rng.uniform(-1, 1, len(ci))
    return rp.astype(np.int32), ci, va


def scattered(n, per_row, seed):
    rng = np.random.default_rng(seed)
    ci = np.sort(rng.integers(0, n, (n, per_row), dtype=np.int64), axis=1)
    rp = (np.arange(n + 1, dtype=np.int64) * per_row)
    return rp.astype(np.int32), ci.ravel().astype(np.int32), rng.uniform(-1, 1, n * per_row)


def bench(name, rp, ci, va, fmts=("csr", "ell", "hyb")):
    n, nnz = len(rp) - 1, len(ci)
    x = ra.LocalVector(data=np.random.default_rng(0).uniform(-1, 1, n)); y = ra.LocalVector(); y.Allocate("", n)
    for fmt in fmts:
        A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
        if fmt != "csr" and A.ConvertTo({"ell": ra.ELL, "hyb": ra.HYB}[fmt]) == ra.CSR:
            print("%-26s %-4s refused" % (name, fmt)); continue
        for _ in range(3):
            A.Apply(x, y)
        capi.check(lib.ramd_timer_start())
        reps = 20
        for _ in range(reps):
            A.Apply(x, y)
        ms = C.c_double(0); capi.check(lib.ramd_timer_stop(C.byref(ms)))
        t = ms.value / reps
        B = 4 * (n + nnz) + 8 * (2 * n + nnz)
        print("%-26s %-4s n=%9d nnz/row=%5.1f  %8.3f ms  %7.1f GB/s (CSR bytes)" % (name, fmt, n, nnz / n, t, B / t / 1e6), flush=True)


which = sys.argv[1:] or ["p7", "s27", "band35", "scatter"]
if "p7" in which:
    from rocalution_amd import generators as gen
    bench("poisson7 192^3", *gen.poisson7(192))
if "s27" in which:
    bench("stencil27 160^3", *stencil27(160))
if "band35" in which:
    bench("banded 35/row, hb 4000", *banded(6_000_000, 35, 4000, 1))
    bench("banded 35/row, hb 400000", *banded(6_000_000, 35, 400000, 2))
if "scatter" in which:
    bench("scattered 16/row", *scattered(8_000_000, 16, 3))
