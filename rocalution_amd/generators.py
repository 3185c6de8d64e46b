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


def laplace27(nx, ny=None, nz=None, dtype=np.float64):
    """the reference's own 3-D operator (clients/include/utility.hpp:110-177 ``gen_3d_laplacian``; cubes there): 27-point stencil,
    row ``r = (z*ny + y)*nx + x``, 26 on the diagonal, -1 at every neighbour of the 3 x 3 x 3 box the lattice has, ascending
    columns.  The device generator is LocalMatrix.GenLaplace27 (csrc/matrix.hip k_lap27_fill)."""
    ny = nx if ny is None else ny
    nz = nx if nz is None else nz
    n = nx * ny * nz
    r = np.arange(n, dtype=np.int64)
    x, y, z = r % nx, (r // nx) % ny, r // (nx * ny)
    cols, mask = [], []
    for sz in (-1, 0, 1):
        for sy in (-1, 0, 1):
            for sx in (-1, 0, 1):
                cols.append(r + (sz * ny + sy) * nx + sx)
                mask.append((x + sx >= 0) & (x + sx < nx) & (y + sy >= 0) & (y + sy < ny) & (z + sz >= 0) & (z + sz < nz))
    cols, mask = np.stack(cols, axis=1), np.stack(mask, axis=1)
    vals = np.where(cols == r[:, None], 26.0, -1.0).astype(dtype)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(mask.sum(axis=1), out=rp[1:])
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


# ---------------------------------------------------------------------------------------------
# af_shell10-class surrogate (BASELINE.json config 3).  SuiteSparse Schenk_AFE/af_shell10 is a
# sheet-metal-forming shell model: n = 1 508 065 = 5 x 301 613 (5 unknowns per mesh node),
# 52 259 885 non-zeros after symmetric expansion = 34.65 per row, SPD.  There is no network on the
# benchmark box, so the file cannot be fetched; this generator produces a matrix of the same class:
#   * a 2-D nx x ny node mesh of quadrilateral cells, every cell split into two triangles along one
#     of its two diagonals chosen by an integer hash  ->  node degree 4..8 (mean 6), i.e. 25..45
#     entries per row (mean 35) and an irregular dependency DAG for the triangular solves;
#   * 5 unknowns per node, dense symmetric 5x5 coupling blocks;
#   * A = sum_e (edge Laplacian (x) K_e) + diag(shift*(1+a)) (a = unknown within the node) with K_e symmetric, strictly diagonally dominant
#     -> A is symmetric positive definite.
# All values are multiples of 1/128 below 2^8, so every sum is exact and the matrix does not depend
# on the accumulation order.  shell_surrogate(549, 549) gives n = 1 507 005, nnz = 52 635 425 (34.93 per row).
_SHELL_DOF = 5


def _mix64(a):
    """splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)"""
    a = (a ^ (a >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    a = (a ^ (a >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return a ^ (a >> np.uint64(31))


def _edge_hash(i, j, kind, seed):
    with np.errstate(over="ignore"):
        k = (i.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + j.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
             + np.uint64(kind) * np.uint64(0x165667B19E3779F9) + np.uint64(seed))
        return _mix64(_mix64(k))


def shell_surrogate(nx, ny=None, seed=1, shift=1.0 / 64, dtype=np.float64):
    """CSR arrays (int32 offsets/columns, sorted rows) of the af_shell10-class surrogate."""
    ny = nx if ny is None else ny
    D = _SHELL_DOF
    nn = nx * ny
    node = np.arange(nn, dtype=np.int64)
    i, j = node % nx, node // nx
    # diagonal of cell (i, j) [lower-left node]: bit 0 of its hash: 0 -> (i,j)-(i+1,j+1), 1 -> (i+1,j)-(i,j+1)
    def cell_diag(ci, cj):
        ok = (ci >= 0) & (ci < nx - 1) & (cj >= 0) & (cj < ny - 1)
        return ok, (_edge_hash(np.where(ok, ci, 0), np.where(ok, cj, 0), 7, seed) & np.uint64(1)).astype(np.int64)
    # the 9 neighbour slots in ascending node-id order; each edge is hashed at its lower (smaller id) end
    slots = []
    for dj in (-1, 0, 1):
        for di in (-1, 0, 1):
            ii, jj = i + di, j + dj
            inside = (ii >= 0) & (ii < nx) & (jj >= 0) & (jj < ny)
            if di == 0 and dj == 0:
                slots.append((di, dj, np.ones(nn, bool), None))
                continue
            # anchor = lower end of the edge; kind: 0 horizontal, 1 vertical, 2 diagonal "/", 3 diagonal "\"
            lo_i, lo_j = np.where(dj < 0, ii, np.where(dj > 0, i, np.minimum(i, ii))), np.minimum(j, jj)
            if dj == 0:
                kind, present = 0, inside
            elif di == 0:
                kind, present = 1, inside
            else:
                # cell containing the diagonal: lower-left corner (min i, min j)
                ci, cj = np.minimum(i, ii), np.minimum(j, jj)
                ok, bit = cell_diag(ci, cj)
                up_right = (di * dj) > 0  # "/" diagonal connects (ci,cj)-(ci+1,cj+1)
                kind = 2 if up_right else 3
                present = inside & ok & (bit == (0 if up_right else 1))
            h = _edge_hash(np.where(present, lo_i, 0), np.where(present, lo_j, 0), kind, seed)
            slots.append((di, dj, present, h))
    a_idx, b_idx = np.meshgrid(np.arange(D), np.arange(D), indexing="ij")
    pair_bit = np.zeros((D, D), dtype=np.uint64)
    k = 0
    for a in range(D):
        for b in range(a + 1, D):
            pair_bit[a, b] = pair_bit[b, a] = 8 + k
            k += 1
    blocks = np.zeros((nn, 9, D, D), dtype=np.float64)
    mask = np.zeros((nn, 9), dtype=bool)
    diag = np.zeros((nn, D, D), dtype=np.float64)
    for s, (di, dj, present, h) in enumerate(slots):
        mask[:, s] = present
        if h is None:
            continue
        c = (16.0 + (h & np.uint64(15)).astype(np.float64)) / 16.0  # edge stiffness in [1, 2)
        sgn = 1.0 - 2.0 * ((h[:, None, None] >> pair_bit[None]) & np.uint64(1)).astype(np.float64)
        K = np.where(a_idx[None] == b_idx[None], 1.0, sgn / 8.0) * c[:, None, None]
        K = np.where(present[:, None, None], K, 0.0)
        blocks[:, s] = -K
        diag += K
    diag += np.diag(shift * (1.0 + np.arange(D)))[None]  # per-unknown shift: the constant vector is no eigenvector
    blocks[:, 4] = diag
    # expand to scalar rows: row (node, a) holds for every present slot s the entries (s, b), b = 0..D-1
    nbr = np.stack([node + dj * nx + di for (di, dj, _, _) in slots], axis=1)  # [nn, 9]
    cols = (nbr[:, None, :, None] * D + np.arange(D)[None, None, None, :])  # [nn, 1, 9, D]
    cols = np.broadcast_to(cols, (nn, D, 9, D))
    vals = np.transpose(blocks, (0, 2, 1, 3))  # [nn, a, s, b]
    m = np.broadcast_to(mask[:, None, :, None], (nn, D, 9, D))
    counts = np.repeat(mask.sum(axis=1) * D, D)
    rp = np.zeros(nn * D + 1, dtype=np.int64)
    np.cumsum(counts, out=rp[1:])
    assert rp[-1] < 2 ** 31
    return rp.astype(np.int32), cols[m].astype(np.int32), np.ascontiguousarray(vals[m]).astype(dtype)


def write_mtx_symmetric(path, rp, ci, va):
    """MatrixMarket `coordinate real symmetric` file holding the lower triangle (col <= row) of a
    symmetric CSR matrix -- the storage af_shell10.mtx itself uses; the reader's symmetric expansion
    (src/base/host/host_io.cpp:218-272) restores the full pattern."""
    n = len(rp) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    keep = ci <= rows
    r1, c1, v1 = rows[keep] + 1, ci[keep].astype(np.int64) + 1, va[keep]
    with open(path, "wb") as f:
        f.write(b"%%MatrixMarket matrix coordinate real symmetric\n")
        f.write(("%d %d %d\n" % (n, n, len(v1))).encode())
        try:
            import pyarrow as pa
            import pyarrow.csv as pacsv
            tbl = pa.table({"r": r1, "c": c1, "v": v1.astype(np.float64)})
            pacsv.write_csv(tbl, f, pacsv.WriteOptions(include_header=False, delimiter=" "))
        except ImportError:
            np.savetxt(f, np.column_stack([r1, c1, v1]), fmt="%d %d %.17g")
    return len(v1)


# ---------------------------------------------------------------------------------------------
# Variants of the config-3 class: the same kind of operator -- 5 unknowns per mesh node, ~35 entries per
# row, SPD -- with the node NUMBERINGS a real file may carry.  The surrogate above numbers its nodes lexicographically, which
# is exactly what the tile construction of the triangular solves keys on (chains of consecutively numbered, dependent rows);
# SuiteSparse af_shell10 is an unstructured sheet-metal mesh in whatever order the pre-processor left it.
#   "lex"       the surrogate as it is
#   "rcm"       the same mesh, nodes renumbered by reverse Cuthill-McKee (scipy.sparse.csgraph) -- a bandwidth-reducing order,
#               level sets of a breadth-first search from a corner: consecutive nodes are mostly NOT neighbours
#   "delaunay"  a Delaunay triangulation of jittered lattice points (valence 4 .. 9), in reverse Cuthill-McKee order
#   "random"    the surrogate's mesh with a random node permutation (the adversarial case: no locality at all)
#   "morton"    the surrogate's mesh in Z-order (bits of the lattice coordinates interleaved): locality without lines or fronts
def _fe_from_edges(nn, eu, ev, seed, shift=1.0 / 64, dtype=np.float64):
    """A = sum over edges (u, v) of [[K, -K], [-K, K]] (K = K_e: symmetric 5 x 5, diagonally dominant, from a hash of the edge)
    + diag(shift (1 + a)): SPD; all values multiples of 1/128, so every sum is exact"""
    import scipy.sparse as sp
    D = _SHELL_DOF
    lo, hi = np.minimum(eu, ev).astype(np.int64), np.maximum(eu, ev).astype(np.int64)
    h = _edge_hash(lo, hi, 5, seed)
    c = (16.0 + (h & np.uint64(15)).astype(np.float64)) / 16.0
    pair_bit = np.zeros((D, D), dtype=np.uint64)
    k = 0
    for a in range(D):
        for b in range(a + 1, D):
            pair_bit[a, b] = pair_bit[b, a] = 8 + k
            k += 1
    sgn = 1.0 - 2.0 * ((h[:, None, None] >> pair_bit[None]) & np.uint64(1)).astype(np.float64)
    eye = np.eye(D, dtype=bool)
    K = np.where(eye[None], 1.0, sgn / 8.0) * c[:, None, None]  # [ne, D, D]
    a_idx, b_idx = np.meshgrid(np.arange(D), np.arange(D), indexing="ij")

    def block_coo(rn, cn, V):
        r = (rn[:, None, None] * D + a_idx[None]).ravel()
        cc = (cn[:, None, None] * D + b_idx[None]).ravel()
        return r, cc, V.ravel()
    parts = [block_coo(lo, hi, -K), block_coo(hi, lo, -K), block_coo(lo, lo, K), block_coo(hi, hi, K)]
    node = np.arange(nn, dtype=np.int64)
    parts.append(block_coo(node, node, np.broadcast_to(np.diag(shift * (1.0 + np.arange(D)))[None], (nn, D, D)).copy()))
    r = np.concatenate([p[0] for p in parts]); cc = np.concatenate([p[1] for p in parts]); v = np.concatenate([p[2] for p in parts])
    A = sp.coo_matrix((v, (r, cc)), shape=(nn * D, nn * D)).tocsr()
    A.sum_duplicates(); A.sort_indices()
    # (the zero blocks of node pairs do not occur: every stored block belongs to an edge or to a node's diagonal)
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(dtype)


def _permute_nodes(rp, ci, va, perm_new_of_old):
    """P A P^T with whole nodes moved: node k becomes node perm_new_of_old[k], its 5 unknowns stay together and in order"""
    import scipy.sparse as sp
    D = _SHELL_DOF
    n = len(rp) - 1
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    old_of_new = np.empty_like(perm_new_of_old)
    old_of_new[perm_new_of_old] = np.arange(len(perm_new_of_old))
    idx = (old_of_new[:, None] * D + np.arange(D)[None]).ravel()
    B = A[idx][:, idx].tocsr()
    B.sort_indices()
    return B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.astype(va.dtype)


def _node_graph(rp, ci):
    import scipy.sparse as sp
    D = _SHELL_DOF
    n = len(rp) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp)) // D
    G = sp.coo_matrix((np.ones(len(ci), np.int8), (rows, ci.astype(np.int64) // D)), shape=(n // D, n // D)).tocsr()
    G.sum_duplicates()
    return G


def _rcm_new_of_old(G):
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    order = reverse_cuthill_mckee(G.tocsr(), symmetric_mode=True)  # order[new] = old
    new_of_old = np.empty(len(order), dtype=np.int64)
    new_of_old[order] = np.arange(len(order))
    return new_of_old


def shell_variant(nx, kind="lex", seed=1, dtype=np.float64):
    """CSR arrays of a config-3-class operator with nx x nx mesh nodes in the numbering `kind` (see above)"""
    if kind == "lex":
        return shell_surrogate(nx, seed=seed, dtype=dtype)
    if kind in ("rcm", "random", "morton"):
        rp, ci, va = shell_surrogate(nx, seed=seed, dtype=dtype)
        if kind == "rcm":
            p = _rcm_new_of_old(_node_graph(rp, ci))
        elif kind == "morton":
            # Z-order of the node lattice (bits of i and j interleaved): what a space-filling-curve partitioner leaves behind
            node = np.arange(nx * nx, dtype=np.int64)
            i, j = node % nx, node // nx
            code = np.zeros(nx * nx, dtype=np.int64)
            for b in range(int(nx - 1).bit_length()):
                code |= ((i >> b) & 1) << (2 * b)
                code |= ((j >> b) & 1) << (2 * b + 1)
            p = np.empty(nx * nx, dtype=np.int64)
            p[np.argsort(code, kind="stable")] = node  # new number of every old node
        else:
            p = np.random.default_rng(seed).permutation(nx * nx).astype(np.int64)
        return _permute_nodes(rp, ci, va, p)
    if kind == "delaunay":
        from scipy.spatial import Delaunay
        rng = np.random.default_rng(seed)
        gx, gy = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(nx, dtype=np.float64), indexing="xy")
        pts = np.column_stack([gx.ravel(), gy.ravel()]) + rng.uniform(-0.35, 0.35, (nx * nx, 2))
        tri = Delaunay(pts).simplices
        e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [0, 2]]])
        e = np.unique(np.sort(e, axis=1), axis=0)
        # (the hull of a jittered lattice has a few long sliver edges: keep edges of at most 2.5 lattice spacings)
        keep = np.linalg.norm(pts[e[:, 0]] - pts[e[:, 1]], axis=1) <= 2.5
        e = e[keep]
        rp, ci, va = _fe_from_edges(nx * nx, e[:, 0], e[:, 1], seed, dtype=dtype)
        return _permute_nodes(rp, ci, va, _rcm_new_of_old(_node_graph(rp, ci)))
    raise ValueError(kind)
