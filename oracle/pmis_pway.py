"""TEST INFRASTRUCTURE (not product code): the reference's PMIS aggregation restated on the CPU, in its single-process form
and in a P-way SERIAL mode -- P ranks emulated one after the other in this process, every message of the reference's
algorithm an explicit copy between their arrays.  Pure-Python loops: small matrices only.

What it follows (paths relative to /root/reference, rocALUTION 3.2.0):
  * the driver   src/base/global_matrix.cpp:2647-3121  GlobalMatrix::AMGPMISAggregate  (with one process it calls the
    LocalMatrix routine, :2669-2675, whose kernels are the same with an empty ghost part);
  * the kernels  src/base/host/host_matrix_csr.cpp
        AMGComputeStrongConnections :5098-5160      hash1 :5162-5168      AMGPMISInitializeState :5171-5222
        lexographical_max :5224-5240                AMGBoundaryNnz :4941-5006 / AMGExtractBoundary :5009-5080
        AMGExtractBoundaryState :5253-5337          AMGPMISFindMaxNeighbourNode :5340-5533
        AMGPMISAddUnassignedNodesToAggregations :5536-5640   AMGPMISInitializeAggregateGlobalIndices :5642-5660
  * the decomposition (rows in contiguous blocks, ghost columns numbered by ascending global index = by owner, the
    boundary index per neighbour)  clients/include/common.hpp:55-431.

How it is pinned: `pmis_single` reproduces the (connections, aggregates, root nodes) the genuine rocALUTION host backend
returns (tests/golden/*: the arrays `test_amg_pmis_aggregation_vs_golden` checks the device kernels against) bit for bit;
`pmis_pway` has no reference run to compare with -- this image has no MPI -- and is anchored on the property its design
rests on: its result is the single-process result for every P and every block boundary (equal hashes of two nodes within
two hops aside, where the visiting order decides; hash1 halves a 32-bit mix, so such pairs exist but are rare at test
sizes).  tests/test_cpu_host.py checks both; the `-m gpu` suite checks the device implementation against this module.
"""
import numpy as np


def hash1(x):
    """host_matrix_csr.cpp:5162-5168 (32-bit unsigned arithmetic)"""
    x &= 0xFFFFFFFF
    x = (((x >> 16) ^ x) * 0x45D9F3B) & 0xFFFFFFFF
    x = (((x >> 16) ^ x) * 0x45D9F3B) & 0xFFFFFFFF
    x = ((x >> 16) ^ x) & 0xFFFFFFFF
    return x // 2


def _lexmax(ti, tj):
    """lexographical_max(&ti, &tj) :5224-5240: tj when it is larger in (s, v), else ti"""
    if tj[0] > ti[0]:
        return tj
    if tj[0] == ti[0] and tj[1] > ti[1]:
        return tj
    return ti


class _Rank:
    """one rank's share: interior / ghost CSR, the ghost-to-global map, the halo pattern"""

    def __init__(self, rp, ci, va, offsets, r):
        lo, hi = int(offsets[r]), int(offsets[r + 1])
        self.lo, self.hi, self.n = lo, hi, hi - lo
        irp, ici, iva, grp, gcols, gva = [0], [], [], [0], [], []
        for i in range(lo, hi):
            for j in range(rp[i], rp[i + 1]):
                c = int(ci[j])
                if lo <= c < hi:
                    ici.append(c - lo); iva.append(va[j])
                else:
                    gcols.append(c); gva.append(va[j])
            irp.append(len(ici)); grp.append(len(gcols))
        self.l2g = sorted(set(gcols))  # ghost column j <-> global column l2g[j]
        pos = {g: k for k, g in enumerate(self.l2g)}
        self.irp, self.ici, self.iva = irp, ici, iva
        self.grp, self.gci, self.gva = grp, [pos[g] for g in gcols], gva
        self.ng = len(self.l2g)
        self.nnz = len(ici)


def _exchange(ranks, arrays):
    """CommunicateAsync_/Sync_ + SetContinuousValues: the entries [n, n + ng) of every rank's array from the owners"""
    owner = {}
    for q, R in enumerate(ranks):
        for i in range(R.n):
            owner[R.lo + i] = (q, i)
    for R, a in zip(ranks, arrays):
        for k, g in enumerate(R.l2g):
            q, i = owner[g]
            a[R.n + k] = arrays[q][i]


def _connections(R, diag, eps):
    """:5098-5160 -- conn[j] over the interior entries, conn[nnz + j] over the ghost entries"""
    eps2 = eps * eps
    conn = [False] * (R.nnz + len(R.gci))
    for i in range(R.n):
        e = eps2 * diag[i]
        for j in range(R.irp[i], R.irp[i + 1]):
            c, v = R.ici[j], R.iva[j]
            conn[j] = (c != i) and (v * v > e * diag[c])
        for j in range(R.grp[i], R.grp[i + 1]):
            c, v = R.gci[j], R.gva[j]
            conn[R.nnz + j] = bool(v * v > e * diag[R.n + c])
    return conn


def _strong(R, conn, i):
    """strongly connected neighbours of row i in the reference's visiting order: (extended index, global number)"""
    out = []
    for j in range(R.irp[i], R.irp[i + 1]):
        if conn[j]:
            out.append((R.ici[j], R.lo + R.ici[j]))
    for j in range(R.grp[i], R.grp[i + 1]):
        if conn[R.nnz + j]:
            out.append((R.n + R.gci[j], R.l2g[R.gci[j]]))
    return out


def pmis_pway(rp, ci, va, eps, offsets):
    """GlobalMatrix::AMGPMISAggregate over len(offsets) - 1 emulated ranks.  Returns (aggregates, root nodes) of all rows
    in global order: the global aggregate number (-2: no strong connection) and the global number of its root node."""
    P = len(offsets) - 1
    ranks = [_Rank(rp, ci, va, offsets, r) for r in range(P)]
    # :2768-2780 diagonal (ExtractDiagonal: zero where the row stores none) and its halo update
    diag = []
    for R in ranks:
        d = [0.0] * (R.n + R.ng)
        for i in range(R.n):
            for j in range(R.irp[i], R.irp[i + 1]):
                if R.ici[j] == i:
                    d[i] = R.iva[j]
                    break
        diag.append(d)
    _exchange(ranks, diag)
    conn = [_connections(R, d, eps) for R, d in zip(ranks, diag)]
    # :2787-2815 state and hash, with their halo updates
    max_state, hashv = [], []
    for R, cn in zip(ranks, conn):
        ms, hv = [0] * (R.n + R.ng), [0] * (R.n + R.ng)
        for i in range(R.n):
            ms[i] = 0 if _strong(R, cn, i) else -2
            hv[i] = hash1(i + R.lo)
        max_state.append(ms); hashv.append(hv)
    _exchange(ranks, max_state)
    _exchange(ranks, hashv)
    # :2817-2886 the boundary rows' strong neighbours (global columns), shipped once: ext[q][g] for ghost node g of rank q
    owner = {}
    for q, R in enumerate(ranks):
        for i in range(R.n):
            owner[R.lo + i] = (q, i)
    agg = [[0] * (R.n + R.ng) for R in ranks]
    while True:
        state = [list(ms) for ms in max_state]
        # :2931-2954 AMGExtractBoundaryState + CommunicateCSR: per ghost node the (state, hash, global column) LIST
        ext = []
        for R in ranks:
            lists = []
            for g in R.l2g:
                q, i = owner[g]
                Q = ranks[q]
                lists.append([(max_state[q][e], hashv[q][e], gc) for e, gc in _strong(Q, conn[q], i)])
            ext.append(lists)
        undecided = False
        for r, R in enumerate(ranks):  # :2957-2970 AMGPMISFindMaxNeighbourNode
            st, hv = state[r], hashv[r]
            for i in range(R.n):
                t = (st[i], hv[i], i)
                for e, _ in _strong(R, conn[r], i):
                    t = _lexmax((st[e], hv[e], e), t)
                if t[2] < R.n:
                    for e, _ in _strong(R, conn[r], t[2]):
                        t = _lexmax((st[e], hv[e], e), t)
                else:
                    for s, v, gc in ext[r][t[2] - R.n]:
                        t = _lexmax((s, v, gc - R.lo if R.lo <= gc < R.hi else -1), t)
                if st[i] == 0:
                    if t[2] == i:
                        max_state[r][i] = 1
                        agg[r][i] = 1
                    elif t[0] == 1:
                        max_state[r][i] = -1
                        agg[r][i] = 0
                    else:
                        undecided = True
        _exchange(ranks, max_state)
        if not undecided:  # (:2985-2997 the all-reduce of the flag)
            break
    # :3013-3029 root nodes; :3031-3060 aggregate numbers = exclusive scan, shifted by the ranks before
    roots = []
    for r, R in enumerate(ranks):
        rt = [-1] * (R.n + R.ng)
        for i in range(R.n):
            rt[i] = R.lo + i if agg[r][i] == 1 else -1
        roots.append(rt)
    _exchange(ranks, roots)
    before = 0
    for r, R in enumerate(ranks):
        s = 0
        for i in range(R.n):
            t = agg[r][i]
            agg[r][i] = s + before
            s += t
        before += s
    _exchange(ranks, agg)
    for _ in range(2):  # :3062-3105 AMGPMISAddUnassignedNodesToAggregations, twice, with the halo updates
        state = [list(ms) for ms in max_state]
        for r, R in enumerate(ranks):
            st = state[r]
            for i in range(R.n):
                if st[i] == -1:
                    gcol = -1
                    for j in range(R.irp[i], R.irp[i + 1]):
                        if conn[r][j] and st[R.ici[j]] == 1:
                            c = R.ici[j]
                            agg[r][i] = agg[r][c]; max_state[r][i] = 1; roots[r][i] = roots[r][c]
                            gcol = R.lo + c
                            break
                    for j in range(R.grp[i], R.grp[i + 1]):
                        if conn[r][R.nnz + j] and st[R.n + R.gci[j]] == 1:
                            c = R.gci[j]
                            if gcol == -1 or (gcol >= 0 and R.l2g[c] < gcol):
                                agg[r][i] = agg[r][R.n + c]; max_state[r][i] = 1; roots[r][i] = roots[r][R.n + c]
                                break
                elif st[i] == -2:
                    agg[r][i] = -2
        _exchange(ranks, agg)
        _exchange(ranks, roots)
        _exchange(ranks, max_state)
    A = np.concatenate([np.asarray(agg[r][:R.n], dtype=np.int64) for r, R in enumerate(ranks)])
    Rt = np.concatenate([np.asarray(roots[r][:R.n], dtype=np.int64) for r, R in enumerate(ranks)])
    return A, Rt


def pmis_single(rp, ci, va, eps):
    """the single-process form: (connections per entry, aggregates, root nodes)"""
    n = len(rp) - 1
    A, Rt = pmis_pway(rp, ci, va, eps, [0, n])
    R = _Rank(rp, ci, va, [0, n], 0)
    d = [0.0] * n
    for i in range(n):
        for j in range(rp[i], rp[i + 1]):
            if ci[j] == i:
                d[i] = va[j]
                break
    return np.asarray(_connections(R, d, eps), dtype=np.int32), A, Rt
