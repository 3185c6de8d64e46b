"""Row-block domain decomposition: host-side setup logic and communicator bootstrap.

Behaviour of the reference's distribution helper (clients/include/common.hpp:55-431 ``distribute_matrix``):
contiguous row blocks (``local = nrow // P``, the first ``nrow % P`` ranks get one more, :92-113), local
columns -> *interior* matrix, remote columns -> *ghost* matrix whose column j is position j of a compact
receive buffer, ``boundary_index`` = local rows whose values a neighbour needs, per neighbour.
Written from scratch on NumPy; the exchange of "who needs what" goes through torch.distributed
(``all_gather_object``), so it works with the gloo backend on CPU-only machines too.

One process per GPU.  The DATA plane (halo exchange, scalar all-reduce) is RCCL inside the library
(``ramd_comm_init_rccl``); torch.distributed is only the control plane that ships the 128-byte id.
A host-staged callback transport (``make_callback_comm``) exists for tests that run several ranks
on one GPU and as a fallback.
"""
import ctypes as C

import numpy as np


def partition_rows(nrow, nproc):
    """-> offsets[nproc+1]; rank r owns rows [offsets[r], offsets[r+1])  (common.hpp:92-113)"""
    base, rem = divmod(int(nrow), int(nproc))
    sizes = [base + (1 if r < rem else 0) for r in range(nproc)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def owner_of(cols, offsets):
    return (np.searchsorted(offsets, cols, side="right") - 1).astype(np.int64)


def split_rows(rp, ci, va, offsets, rank):
    """Local pieces of a global CSR matrix for `rank`.

    Returns dict(interior=(rp,ci,va), ghost=(rp,ci,va), recv_peers, recv_offset, recv_global)
    where ghost columns index the receive buffer: neighbours in ascending rank order, each
    neighbour's columns in ascending global index.
    """
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    nloc = hi - lo
    s, e = int(rp[lo]), int(rp[hi])
    cols = np.asarray(ci[s:e], dtype=np.int64)
    vals = np.asarray(va[s:e])
    rows = np.repeat(np.arange(nloc, dtype=np.int64), np.diff(rp[lo:hi + 1]))
    local = (cols >= lo) & (cols < hi)
    # interior
    irp = np.zeros(nloc + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows[local], minlength=nloc), out=irp[1:])
    ici = (cols[local] - lo).astype(np.int32)
    iva = vals[local].copy()
    # ghost: receive-buffer numbering
    gcols = cols[~local]
    uniq = np.unique(gcols)  # ascending global index == (owner rank, index) order for row blocks
    owners = owner_of(uniq, offsets)
    peers = np.unique(owners)
    recv_offset = np.concatenate([[0], np.cumsum([np.sum(owners == p) for p in peers])]).astype(np.int64)
    grp = np.zeros(nloc + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows[~local], minlength=nloc), out=grp[1:])
    gci = np.searchsorted(uniq, gcols).astype(np.int32)
    gva = vals[~local].copy()
    return dict(interior=(irp, ici, iva), ghost=(grp, gci, gva), recv_peers=peers.astype(np.int32),
                recv_offset=recv_offset, recv_global=uniq, row_begin=lo, row_end=hi)


def build_halo_plan(piece, offsets, rank, all_gather_object):
    """Complete a rank's piece with the SEND side: which of my rows every neighbour needs.

    `all_gather_object(obj) -> list` is the only collective used (torch.distributed.all_gather_object
    or a single-process stand-in).  Returns dict(peers, send_offset, recv_offset, boundary_index)."""
    lo = int(offsets[rank])
    needs = {}
    for k, p in enumerate(piece["recv_peers"]):
        needs[int(p)] = piece["recv_global"][piece["recv_offset"][k]:piece["recv_offset"][k + 1]]
    everyone = all_gather_object(needs)  # everyone[q][p] = global rows q wants from p
    send_peers, boundary = [], []
    send_offset = [0]
    for q, wants in enumerate(everyone):
        if q != rank and rank in wants and len(wants[rank]):
            send_peers.append(q)
            boundary.append(np.asarray(wants[rank], dtype=np.int64) - lo)
            send_offset.append(send_offset[-1] + len(wants[rank]))
    recv_peers = [int(p) for p in piece["recv_peers"]]
    if send_peers != recv_peers:
        raise ValueError("non-symmetric communication pattern (send peers %r, recv peers %r): this "
                         "implementation pairs every send with a receive" % (send_peers, recv_peers))
    bidx = np.concatenate(boundary).astype(np.int32) if boundary else np.zeros(0, np.int32)
    return dict(peers=np.asarray(send_peers, dtype=np.int32), send_offset=np.asarray(send_offset, np.int32),
                recv_offset=piece["recv_offset"].astype(np.int32), boundary_index=bidx)


# --------------------------------------------------------------------------- communicators
def init_rccl_comm(rank, world, dist):
    """RCCL communicator inside the library; `dist` (torch.distributed, any backend) ships the id."""
    import torch
    from . import capi
    lib = capi.load()
    uid = C.create_string_buffer(128)
    if rank == 0:
        capi.check(lib.ramd_comm_unique_id(uid))
    t = torch.tensor(list(uid.raw), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    uid = C.create_string_buffer(bytes(t.tolist()), 128)
    comm = C.c_void_p()
    capi.check(lib.ramd_comm_init_rccl(rank, world, uid, C.byref(comm)))
    return comm


_EXCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int64),
                    C.c_void_p, C.POINTER(C.c_int64))
_ARED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
_keepalive = []


def make_callback_comm(rank, world, dist):
    """Host-staged transport: the library hands pinned host buffers to these callbacks, which move
    them with torch.distributed point-to-point ops (gloo)."""
    import torch
    from . import capi
    lib = capi.load()

    def exchange(user, npeers, peers, send_host, send_off, recv_host, recv_off):
        try:
            ops, keep = [], []
            for k in range(npeers):
                p = int(peers[k])
                ns, nr = send_off[k + 1] - send_off[k], recv_off[k + 1] - recv_off[k]
                sbuf = (C.c_uint8 * ns).from_address(send_host + send_off[k]) if ns else None
                rbuf = (C.c_uint8 * nr).from_address(recv_host + recv_off[k]) if nr else None
                if ns:
                    ts = torch.frombuffer(sbuf, dtype=torch.uint8)
                    ops.append(dist.P2POp(dist.isend, ts, p))
                    keep.append(ts)
                if nr:
                    tr = torch.frombuffer(rbuf, dtype=torch.uint8)
                    ops.append(dist.P2POp(dist.irecv, tr, p))
                    keep.append(tr)
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            return 0
        except Exception as e:  # pragma: no cover
            print("halo exchange callback failed:", repr(e), flush=True)
            return 1

    def allreduce(user, values, count):
        try:
            t = torch.tensor([values[i] for i in range(count)], dtype=torch.float64)
            dist.all_reduce(t)
            for i in range(count):
                values[i] = float(t[i])
            return 0
        except Exception as e:  # pragma: no cover
            print("allreduce callback failed:", repr(e), flush=True)
            return 1

    ex, ar = _EXCH(exchange), _ARED(allreduce)
    _keepalive.extend([ex, ar])
    comm = C.c_void_p()
    capi.check(lib.ramd_comm_init_callback(rank, world, C.cast(ex, C.c_void_p), C.cast(ar, C.c_void_p), None,
                                           C.byref(comm)))
    return comm


class DistributedSolver:
    """Solver<GlobalMatrix,GlobalVector> on one rank (ramd_gsolver_* of the C ABI)."""

    def __init__(self, comm, solver, precond, mixed=False):
        """mixed=True: MixedPrecisionDC (fp64 outer) around an fp32 `solver` with `precond` (none/Jacobi)"""
        from . import capi
        self._capi, self._lib = capi, capi.load()
        self._g = C.c_void_p()
        if mixed:
            capi.check(self._lib.ramd_gsolver_create_mixed(comm, int(solver), int(precond), C.byref(self._g)))
        else:
            capi.check(self._lib.ramd_gsolver_create(comm, int(solver), int(precond), C.byref(self._g)))

    def init_inner(self, abs_tol, rel_tol, div_tol, max_iter):
        self._capi.check(self._lib.ramd_gsolver_init_inner(self._g, abs_tol, rel_tol, div_tol, max_iter))

    def __del__(self):
        try:
            if self._g:
                self._lib.ramd_gsolver_destroy(self._g)
                self._g = None
        except Exception:
            pass

    def setup_poisson(self, N, z_begin, z_end):
        self._capi.check(self._lib.ramd_gsolver_setup_poisson(self._g, N, z_begin, z_end))

    def setup_laplace27(self, N, z_begin, z_end):
        """planes [z_begin, z_end) of the reference's 27-point Laplacian N^3 (clients/include/common.hpp:926-1249), generated on the device"""
        self._capi.check(self._lib.ramd_gsolver_setup_laplace27(self._g, N, z_begin, z_end))

    def setup_csr(self, global_nrow, piece, plan):
        irp, ici, iva = piece["interior"]
        grp, gci, gva = piece["ghost"]
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        arrs = [i32(irp), i32(ici), f64(iva), i32(grp), i32(gci), f64(gva), i32(plan["peers"]),
                i32(plan["send_offset"]), i32(plan["recv_offset"]), i32(plan["boundary_index"])]
        p = [a.ctypes.data_as(C.c_void_p) for a in arrs]
        self._capi.check(self._lib.ramd_gsolver_setup_csr(
            self._g, int(global_nrow), len(irp) - 1, len(ici), p[0], p[1], p[2], len(gci), p[3], p[4], p[5],
            len(plan["peers"]), p[6], p[7], p[8], p[9]))

    def convert(self, fmt):
        self._capi.check(self._lib.ramd_gsolver_convert(self._g, int(fmt)))

    def init(self, abs_tol, rel_tol, div_tol, max_iter, min_iter=0):
        self._capi.check(self._lib.ramd_gsolver_init(self._g, abs_tol, rel_tol, div_tol, min_iter, max_iter))

    def set_basis(self, m):
        self._capi.check(self._lib.ramd_gsolver_set_basis(self._g, int(m)))

    def build(self):
        self._capi.check(self._lib.ramd_gsolver_build(self._g))

    def apply(self, x_local):
        x_local = np.ascontiguousarray(x_local, dtype=np.float64)
        y = np.zeros_like(x_local)
        self._capi.check(self._lib.ramd_gsolver_apply(self._g, x_local.ctypes.data_as(C.c_void_p),
                                                      y.ctypes.data_as(C.c_void_p)))
        return y

    def solve(self, rhs_local, x0_local):
        x = np.ascontiguousarray(x0_local, dtype=np.float64).copy()
        r = None if rhs_local is None else np.ascontiguousarray(rhs_local, dtype=np.float64)
        self._capi.check(self._lib.ramd_gsolver_solve(self._g, None if r is None else r.ctypes.data_as(C.c_void_p),
                                                      x.ctypes.data_as(C.c_void_p)))
        return x

    def amg_info(self):
        """RAMD_PC_GLOBAL_* after build(): (levels, global rows of the coarsest operator, worst Galerkin defect)"""
        lv, rows, worst = C.c_int(0), C.c_int64(0), C.c_double(0)
        self._capi.check(self._lib.ramd_gsolver_amg_info(self._g, C.byref(lv), C.byref(rows), C.byref(worst)))
        return lv.value, rows.value, worst.value

    def amg_level(self, level):
        """RAMD_PC_GLOBAL_* after build(): (global rows, this rank's entries, || A_level 1 ||) of one level's operator"""
        rows, ent, nrm = C.c_int64(0), C.c_int64(0), C.c_double(0)
        self._capi.check(self._lib.ramd_gsolver_amg_level(self._g, int(level), C.byref(rows), C.byref(ent), C.byref(nrm)))
        return rows.value, ent.value, nrm.value

    def result(self):
        it, st, res = C.c_int(0), C.c_int(0), C.c_double(0)
        self._capi.check(self._lib.ramd_gsolver_result(self._g, C.byref(it), C.byref(st), C.byref(res)))
        return it.value, st.value, res.value
