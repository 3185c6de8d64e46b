"""Worker for the 2-rank tests (spawned by test_cpu_host.py / test_gpu_distributed.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _init(rank, world, initfile):
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    return dist


def _gather_obj(dist, world):
    def f(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    return f


def _matrix(kind):
    from rocalution_amd import generators as gen
    if kind == "poisson":
        return gen.poisson7(12)
    if kind == "gr3030":
        return gen.gr_30_30()
    if kind == "lap27":  # the reference's own 3-D operator (gen_3d_laplacian, clients/include/utility.hpp:110-177), 9 x 8 x 10
        return gen.laplace27(9, 8, 10)
    return gen.random_sparse(500, 5, seed=3)


def _symmetrize_pattern(rp, ci, va):
    import scipy.sparse as sp
    n = len(rp) - 1
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    B = (A + A.T).tocsr()  # symmetric pattern + values (SPD-ish with the dominant diagonal)
    B.sort_indices()
    return B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.astype(np.float64)


def cpu_worker(rank, world, initfile, kind, outdir):
    """CPU (gloo) emulation of GlobalMatrix::Apply + CG<GlobalMatrix> built from the PRODUCT's host
    logic (rocalution_amd.distributed) with the oracle supplying the local kernels."""
    import torch
    from oracle import oracle as orc
    from rocalution_amd import distributed as D
    dist = _init(rank, world, initfile)
    orc.build(); orc.set_threads(1)
    rp, ci, va = _matrix(kind)
    if kind == "random":
        rp, ci, va = _symmetrize_pattern(rp, ci, va)
    n = len(rp) - 1
    off = D.partition_rows(n, world)
    piece = D.split_rows(rp, ci, va, off, rank)
    plan = D.build_halo_plan(piece, off, rank, _gather_obj(dist, world))
    lo, hi = piece["row_begin"], piece["row_end"]
    irp, ici, iva = piece["interior"]
    grp, gci, gva = piece["ghost"]

    def halo(xloc):
        send = xloc[plan["boundary_index"]]
        recv = np.zeros(int(plan["recv_offset"][-1]))
        ops = []
        for k, p in enumerate(plan["peers"]):
            s = torch.from_numpy(np.ascontiguousarray(send[plan["send_offset"][k]:plan["send_offset"][k + 1]]))
            r = torch.from_numpy(recv[plan["recv_offset"][k]:plan["recv_offset"][k + 1]])
            ops += [dist.P2POp(dist.isend, s, int(p)), dist.P2POp(dist.irecv, r, int(p))]
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return recv

    def apply(xloc):
        recv = halo(xloc)
        y = orc.csr_apply(irp, ici, iva, xloc)
        if len(gva):
            y = orc.csr_apply_add(grp, gci, gva, recv, 1.0, y)
        return y

    def gdot(a, b):
        t = torch.tensor([float(orc.dot(a, b))], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t[0])

    x = np.random.default_rng(5).uniform(-1, 1, n)
    y = apply(x[lo:hi])
    # CG + Jacobi, global semantics (cg.cpp:366-446 with GlobalVector reductions)
    b = apply(np.ones(hi - lo))
    dinv = orc.extract_inv_diag(irp, ici, iva)
    xs = np.zeros(hi - lo)
    r = b - apply(xs)
    res0 = np.sqrt(gdot(r, r))
    z = dinv * r
    p = z.copy()
    rho = gdot(r, z)
    it = 0
    while True:
        q = apply(p)
        alpha = rho / gdot(p, q)
        xs = xs + alpha * p
        r = r + (-alpha) * q
        it += 1
        res = np.sqrt(gdot(r, r))
        if res / res0 <= 1e-6 or it >= 500:
            break
        z = dinv * r
        rho_old, rho = rho, gdot(r, z)
        p = (rho / rho_old) * p + z
    np.savez(os.path.join(outdir, "r%d.npz" % rank), lo=lo, hi=hi, y=y, xs=xs, it=it, res=res)
    dist.barrier()
    dist.destroy_process_group()


def gpu_worker(rank, world, initfile, kind, outdir, rccl=False):
    """2 ranks sharing ONE GPU: the real GlobalMatrix / GlobalVector / CG path of the library with the
    host-staged callback transport (RCCL refuses two ranks on one device).
    rccl=True: one GPU per rank, halo exchange and scalar sums over RCCL (ramd_comm_init_rccl) -- the transport of the
    multi-GPU bench; needs `world` visible devices."""
    from rocalution_amd import capi, distributed as D
    import rocalution_amd as ra
    dist = _init(rank, world, initfile)
    ra.init_rocalution(rank if rccl else 0)
    comm = D.init_rccl_comm(rank, world, dist) if rccl else D.make_callback_comm(rank, world, dist)
    out = {}
    if kind in ("poisson_slab", "lap27_slab"):
        N = 12
        z0, z1 = (N * rank) // world, (N * (rank + 1)) // world
        n = N ** 3
        lo, hi = z0 * N * N, z1 * N * N
        g = D.DistributedSolver(comm, capi.SOLVER_CG, capi.PC_JACOBI)
        slab = (lambda gg: gg.setup_laplace27(N, z0, z1)) if kind == "lap27_slab" else (lambda gg: gg.setup_poisson(N, z0, z1))
        slab(g)
    else:
        rp, ci, va = _matrix(kind)
        if kind == "random":
            rp, ci, va = _symmetrize_pattern(rp, ci, va)
        n = len(rp) - 1
        off = D.partition_rows(n, world)
        piece = D.split_rows(rp, ci, va, off, rank)
        plan = D.build_halo_plan(piece, off, rank, _gather_obj(dist, world))
        lo, hi = piece["row_begin"], piece["row_end"]
        g = D.DistributedSolver(comm, capi.SOLVER_CG, capi.PC_JACOBI)
        g.setup_csr(n, piece, plan)
    x = np.random.default_rng(5).uniform(-1, 1, n)
    out["y"] = g.apply(x[lo:hi])
    g.init(1e-15, 1e-6, 1e8, 500)
    g.build()
    out["xs"] = g.solve(None, np.zeros(hi - lo))
    it, st, res = g.result()
    # same system again after converting the interior to ELL (ghost -> COO), BiCGStab + BlockJacobi(MC-SGS)
    g2 = D.DistributedSolver(comm, capi.SOLVER_BICGSTAB, capi.PC_MCSGS)
    if kind in ("poisson_slab", "lap27_slab"):
        slab(g2)
    else:
        g2.setup_csr(n, piece, plan)
    g2.init(1e-15, 1e-6, 1e8, 500)
    g2.build()
    g2.convert(ra.ELL)
    out["y_ell"] = g2.apply(x[lo:hi])
    out["xs2"] = g2.solve(None, np.zeros(hi - lo))
    it2, st2, res2 = g2.result()
    # config-5 shape: fp64 defect correction around fp32 CG+Jacobi, both levels Global
    g3 = D.DistributedSolver(comm, capi.SOLVER_CG, capi.PC_JACOBI, mixed=True)
    if kind in ("poisson_slab", "lap27_slab"):
        slab(g3)
    else:
        g3.setup_csr(n, piece, plan)
    g3.init(1e-15, 1e-6, 1e8, 500)
    g3.init_inner(1e-5, 1e-2, 1e20, 100000)
    g3.build()
    out["xs3"] = g3.solve(None, np.zeros(hi - lo))
    it3, st3, res3 = g3.result()
    # BlockJacobi around the wider local preconditioners: SA-AMG / IC on the interior block under CG (the symmetric
    # operators), ILU(0) / MC-SGS under BiCGStab (the random one)
    its4 = []
    for sk, pk in ([(capi.SOLVER_BICGSTAB, capi.PC_ILU0), (capi.SOLVER_BICGSTAB, capi.PC_MCSGS)] if kind == "random"
                   else [(capi.SOLVER_CG, capi.PC_SAAMG), (capi.SOLVER_CG, capi.PC_IC), (capi.SOLVER_CG, capi.PC_UAAMG)]):
        g4 = D.DistributedSolver(comm, sk, pk)
        if kind in ("poisson_slab", "lap27_slab"):
            slab(g4)
        else:
            g4.setup_csr(n, piece, plan)
        g4.init(1e-15, 1e-8, 1e8, 500)
        g4.build()
        out["xs4_%d" % len(its4)] = g4.solve(None, np.zeros(hi - lo))
        it4, st4, res4 = g4.result()
        its4.append((it4, st4))
    out["its4"] = np.array(its4)
    # P-way BlockJacobi parity legs (checked against the oracle's nblocks = world mode): BiCGStab + BlockJacobi(MC-SGS)
    # with the interior converted to ELL and to HYB after Build (config 4's solver), GMRES(30) + BlockJacobi(ILU(0))
    its5 = []
    for sk, pk, fmt in ((capi.SOLVER_BICGSTAB, capi.PC_MCSGS, ra.ELL), (capi.SOLVER_BICGSTAB, capi.PC_MCSGS, ra.HYB),
                        (capi.SOLVER_GMRES, capi.PC_ILU0, ra.CSR)):
        g5 = D.DistributedSolver(comm, sk, pk)
        if kind in ("poisson_slab", "lap27_slab"):
            slab(g5)
        else:
            g5.setup_csr(n, piece, plan)
        g5.init(1e-15, 1e-6, 1e8, 500)
        g5.build()
        if fmt != ra.CSR:
            g5.convert(fmt)
        out["xs5_%d" % len(its5)] = g5.solve(None, np.zeros(hi - lo))
        it5, st5, res5 = g5.result()
        its5.append((it5, st5, res5))
    out["its5"] = np.array(its5)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), lo=lo, hi=hi, it=it, res=res, st=st, it2=it2, st2=st2,
             res2=res2, it3=it3, st3=st3, res3=res3, **out)
    dist.barrier()
    dist.destroy_process_group()


def amg_worker(rank, world, initfile, kind, outdir, rccl=False):
    """aggregation AMG on the GlobalMatrix (RAMD_PC_GLOBAL_*): CG preconditioned by UA-AMG and by SA-AMG on `world` ranks
    sharing one GPU (callback transport).  kind: "poisson_slab" (24^3, z-slabs) or "gr3030x" (the 2-D 9-point gr_30_30
    pattern repeated on a 120 x 120 grid, general CSR split with an unstructured halo plan)"""
    from rocalution_amd import capi, distributed as D
    import rocalution_amd as ra
    dist = _init(rank, world, initfile)
    ra.init_rocalution(rank if rccl else 0)
    comm = D.init_rccl_comm(rank, world, dist) if rccl else D.make_callback_comm(rank, world, dist)
    out = {}
    for tag, pk in (("ua", capi.PC_GLOBAL_UAAMG), ("sa", capi.PC_GLOBAL_SAAMG)):
        g = D.DistributedSolver(comm, capi.SOLVER_CG, pk)
        if kind.startswith("poisson_slab"):
            N = int(kind[len("poisson_slab"):] or 24)
            z0, z1 = (N * rank) // world, (N * (rank + 1)) // world
            lo, hi = z0 * N * N, z1 * N * N
            g.setup_poisson(N, z0, z1)
        else:
            rp, ci, va = amg_matrix(kind)
            n = len(rp) - 1
            off = D.partition_rows(n, world)
            piece = D.split_rows(rp, ci, va, off, rank)
            plan = D.build_halo_plan(piece, off, rank, _gather_obj(dist, world))
            lo, hi = piece["row_begin"], piece["row_end"]
            g.setup_csr(n, piece, plan)
        g.init(1e-15, 1e-8, 1e8, 200)
        g.build()
        out["info_" + tag] = np.array(g.amg_info(), dtype=np.float64)
        out["levels_" + tag] = np.array([g.amg_level(l) for l in range(int(out["info_" + tag][0]))], dtype=np.float64)
        out["x_" + tag] = g.solve(None, np.zeros(hi - lo))
        out["res_" + tag] = np.array(g.result(), dtype=np.float64)
    np.savez(os.path.join(outdir, "r%d.npz" % rank), lo=lo, hi=hi, **out)
    dist.barrier()
    dist.destroy_process_group()


def aggregate_worker(rank, world, initfile, kind, outdir):
    """the distributed aggregation through the C ABI itself (ramd_mat_merge_columns, ramd_mat_amg_pmis_aggregate_global,
    ramd_mat_amg_prolong_global) on the row blocks of a golden matrix; kind = "<golden name>:<even|uneven>" """
    import ctypes as C
    from conftest import load_golden
    from rocalution_amd import capi, distributed as D
    import rocalution_amd as ra
    dist = _init(rank, world, initfile)
    ra.init_rocalution(0)
    lib = capi.load()
    comm = D.make_callback_comm(rank, world, dist)
    name, how = kind.split(":")
    if name == "iso":
        rp, ci, va = amg_matrix(name)
    else:
        g = load_golden(name)
        rp, ci, va = g["rowptr"], g["col"], g["val"]
    n = len(rp) - 1
    off = D.partition_rows(n, world)
    if how == "uneven":  # blocks of very different sizes (the first one two rows)
        cuts = sorted(set([2] + [int(n * f) for f in (0.11, 0.16, 0.55, 0.8, 0.93, 0.97)]))[:world - 1]
        off = np.array([0] + cuts + [n], dtype=np.int64)
    piece = D.split_rows(rp, ci, va, off, rank)
    plan = D.build_halo_plan(piece, off, rank, _gather_obj(dist, world))
    lo, hi = piece["row_begin"], piece["row_end"]
    nloc, nrecv = hi - lo, int(plan["recv_offset"][-1])
    Ai, Ag, blk, Pu, Ps = (ra.LocalMatrix() for _ in range(5))
    Ai.SetDataPtrCSR(*piece["interior"], nrow=nloc, ncol=nloc)
    has_ghost = len(piece["ghost"][1]) > 0
    if has_ghost:
        Ag.SetDataPtrCSR(*piece["ghost"], nrow=nloc, ncol=nrecv)
    capi.check(lib.ramd_mat_merge_columns(Ai._h, Ag._h if has_ghost else None, nrecv, blk._h))
    peers = np.ascontiguousarray(plan["peers"], dtype=np.int32)
    so = np.ascontiguousarray(plan["send_offset"], dtype=np.int64)
    ro = np.ascontiguousarray(plan["recv_offset"], dtype=np.int64)
    bnd = ra.LocalVector(np.int32, data=plan["boundary_index"])
    numbers, conn, agg, roots = (ra.LocalVector(np.int32) for _ in range(4))
    first, mine, total = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    pi32, pi64 = C.POINTER(C.c_int), C.POINTER(C.c_int64)
    capi.check(lib.ramd_mat_amg_pmis_aggregate_global(
        blk._h, 0.01, comm, 0, len(peers), peers.ctypes.data_as(pi32), so.ctypes.data_as(pi64), ro.ctypes.data_as(pi64),
        bnd._h, int(lo), numbers._h, conn._h, agg._h, roots._h, C.byref(first), C.byref(mine), C.byref(total)))
    capi.check(lib.ramd_mat_amg_prolong_global(blk._h, 0, 0.0, 0, conn._h, agg._h, roots._h, total.value, Pu._h))
    capi.check(lib.ramd_mat_amg_prolong_global(blk._h, 1, 2.0 / 3.0, 0, conn._h, agg._h, roots._h, total.value, Ps._h))
    out = dict(lo=lo, hi=hi, nrecv=nrecv, numbers=numbers.CopyToHostData(), agg=agg.CopyToHostData(),
               roots=roots.CopyToHostData(), conn=conn.CopyToHostData(), first=first.value, mine=mine.value,
               total=total.value, recv_global=piece["recv_global"])
    for tag, P in (("pu", Pu), ("ps", Ps)):
        prp, pci, pva = P.CopyToCSR()
        out[tag + "_rp"], out[tag + "_ci"], out[tag + "_va"], out[tag + "_ncol"] = prp, pci, pva, P.GetN()
    # the entries of the block in the reference's layout: conn over [interior entries | ghost entries] of the rank
    brp, bci, _ = blk.CopyToCSR()
    out["blk_rp"], out["blk_ci"] = brp, bci
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def plans_matrices(world, m=600):
    """two operators on world * m rows whose row block of rank 0 couples to no other rank (its exchange plan is the same empty
    one for both), while the other ranks are coupled in a chain -- by one sub-/super-diagonal pair in A, by three in B: the
    boundary pieces, and with them the padded length of the all-gather form, differ between the two on every other rank"""
    import scipy.sparse as sp
    n = world * m
    rng = np.random.default_rng(21)
    def chain(width):
        d = [sp.diags(rng.uniform(-1.0, -0.1, n - k), k) for k in range(1, width + 1)]
        L = sum(d)
        A = (L + L.T).tolil()
        A[:m, m:] = 0.0  # rank 0 stands alone
        A[m:, :m] = 0.0
        A = A.tocsr()
        A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
        A = A.tocsr(); A.sort_indices(); A.eliminate_zeros()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    return chain(1), chain(3)


def plans_worker(rank, world, initfile, kind, outdir):
    """two GlobalMatrix objects on ONE communicator, products with them in alternation (see plans_matrices)"""
    from rocalution_amd import capi, distributed as D
    import rocalution_amd as ra
    dist = _init(rank, world, initfile)
    ra.init_rocalution(0)
    comm = D.make_callback_comm(rank, world, dist)
    mats = plans_matrices(world)
    n = len(mats[0][0]) - 1
    off = D.partition_rows(n, world)
    gs = []
    for rp, ci, va in mats:
        piece = D.split_rows(rp, ci, va, off, rank)
        plan = D.build_halo_plan(piece, off, rank, _gather_obj(dist, world))
        g = D.DistributedSolver(comm, capi.SOLVER_CG, capi.PC_JACOBI)
        g.setup_csr(n, piece, plan)
        gs.append((g, piece, plan))
    lo, hi = gs[0][1]["row_begin"], gs[0][1]["row_end"]
    out = {"npeers": np.array([len(p["peers"]) for _, _, p in gs])}
    rng = np.random.default_rng(9)
    for rep in range(3):
        x = rng.uniform(-1, 1, n)
        for k, (g, _, _) in enumerate(gs):
            out["y%d_%d" % (k, rep)] = g.apply(x[lo:hi])
    # ... and a solve with each (all-reduces between the exchanges)
    for k, (g, _, _) in enumerate(gs):
        g.init(1e-15, 1e-8, 1e8, 500)
        g.build()
        out["xs%d" % k] = g.solve(None, np.zeros(hi - lo))
    np.savez(os.path.join(outdir, "r%d.npz" % rank), lo=lo, hi=hi, **out)
    dist.barrier()
    dist.destroy_process_group()


def amg_matrix(kind):
    import scipy.sparse as sp
    if kind == "gr3030x":
        # 9-point stencil of gr_30_30 (8 on the diagonal, -1 to the 8 neighbours) on a 120 x 120 grid
        N = 120
        t = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
        A = (-sp.kron(t, t) + sp.diags(np.full(N * N, 9.0))).tocsr()
        A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    if kind == "thin":
        # the same stencil on a 16 x 60 grid with random symmetric weights: over 8 ranks a row block is two grid lines thick,
        # so aggregates (two hops) and the columns of the smoothed prolongation (three) reach past the neighbouring rank --
        # the owner of a coarse column is then a rank this one shares no fine halo with
        ny, nx = 16, 60
        rng = np.random.default_rng(33)
        idx = np.arange(ny * nx).reshape(ny, nx)
        rows, cols, vals = [], [], []
        for dy, dx in ((0, 1), (1, -1), (1, 0), (1, 1)):
            a = idx[max(0, -dy):ny - max(0, dy), max(0, -dx):nx - max(0, dx)].ravel()
            b = idx[max(0, dy):ny - max(0, -dy), max(0, dx):nx - max(0, -dx)].ravel()
            w = -rng.uniform(0.5, 1.5, a.size)
            rows += [a, b]; cols += [b, a]; vals += [w, w]
        A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ny * nx, ny * nx)).tocsr()
        A = (A + sp.diags(-np.asarray(A.sum(axis=1)).ravel() + 0.05)).tocsr()
        A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    if kind == "iso":
        # 400 rows: rows 0..99 a component of their own (over 4 even blocks rank 0 has no neighbour), a random symmetric
        # pattern on the rest, and every ninth row with a diagonal so large that none of its couplings is strong (such
        # rows stay outside every aggregate, also where they are boundary rows or ghost nodes)
        n, m = 400, 100
        rng = np.random.default_rng(77)
        def sym(k, dens):
            B = sp.random(k, k, density=dens, random_state=rng, data_rvs=lambda s: -rng.uniform(0.2, 1.0, s))
            B = sp.triu(B, 1)
            return (B + B.T).tocsr()
        band = sp.diags([-np.ones(n - m - 1), -np.ones(n - m - 1)], [-1, 1])  # (keeps the big component connected)
        A = sp.block_diag([sym(m, 0.06) + sp.diags([-np.ones(m - 1), -np.ones(m - 1)], [-1, 1]), sym(n - m, 0.02) + band]).tocsr()
        d = -np.asarray(A.sum(axis=1)).ravel() + 0.1
        d[::9] *= 1e6
        A = (A + sp.diags(d)).tocsr()
        A.sort_indices()
        return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    raise ValueError(kind)


def slab8_worker(rank, world, initfile, kind, outdir):
    """BASELINE.json configs 4 / 5 as they are specified -- "row-split across 8 x MI355X": `world` processes on ONE device over
    the host-staged transport, every rank a z-slab of the Poisson operator (clients/include/common.hpp:92-113 partition rule,
    choreography of src/base/global_matrix.cpp:924-1009).  kind = "<what>:<N>"."""
    from rocalution_amd import capi, distributed as D
    import rocalution_amd as ra
    what, N = kind.split(":")
    N = int(N)
    dist = _init(rank, world, initfile)
    ra.init_rocalution(0)
    comm = D.make_callback_comm(rank, world, dist)
    z0, z1 = (N * rank) // world, (N * (rank + 1)) // world
    lo, hi = z0 * N * N, z1 * N * N
    out = dict(lo=lo, hi=hi)
    if what == "apply":
        g = D.DistributedSolver(comm, capi.SOLVER_CG, capi.PC_JACOBI)
        g.setup_poisson(N, z0, z1)
        y1 = g.apply(np.ones(hi - lo))
        idx = np.arange(lo, hi, dtype=np.int64)
        xv = ((idx * 7 + (idx // N) * 3 + (idx // (N * N)) * 5) % 17 - 8).astype(np.float64)  # small integers: every sum exact
        y2 = g.apply(xv)
        for fmt in (ra.ELL, ra.HYB):
            g.convert(fmt)
            assert np.array_equal(g.apply(xv), y2)
            g.convert(ra.CSR)
        # closed form on this slab: 6 x - the neighbours that exist (the halo planes come from the neighbouring ranks)
        k = np.arange(z0, z1)[:, None, None]; j = np.arange(N)[None, :, None]; i = np.arange(N)[None, None, :]

        def xval(ii, jj, kk):
            r = (kk * N + jj) * N + ii
            return ((r * 7 + (r // N) * 3 + (r // (N * N)) * 5) % 17 - 8).astype(np.float64)
        ref2 = 6.0 * xval(i, j, k)
        nb = np.zeros((z1 - z0, N, N))
        for (di, dj, dk) in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
            ii, jj, kk = i + di, j + dj, k + dk
            ok = (ii >= 0) & (ii < N) & (jj >= 0) & (jj < N) & (kk >= 0) & (kk < N)
            v = xval(np.clip(ii, 0, N - 1), np.clip(jj, 0, N - 1), np.clip(kk, 0, N - 1))
            ref2 = ref2 - np.where(ok, v, 0.0)
            nb = nb + ok
        out["apply_ones_ok"] = bool(np.array_equal(y1, (6.0 - nb).ravel()))
        out["apply_x_ok"] = bool(np.array_equal(y2, np.broadcast_to(ref2, (z1 - z0, N, N)).ravel()))
    else:
        sk, pk, fmt, mixed = {"c4": (capi.SOLVER_BICGSTAB, capi.PC_MCSGS, ra.ELL, False),
                              "c4hyb": (capi.SOLVER_BICGSTAB, capi.PC_MCSGS, ra.HYB, False),
                              "c5": (capi.SOLVER_CG, capi.PC_JACOBI, ra.CSR, True),
                              "gmres": (capi.SOLVER_GMRES, capi.PC_ILU0, ra.CSR, False)}[what]
        g = D.DistributedSolver(comm, sk, pk, mixed=mixed)
        g.setup_poisson(N, z0, z1)
        g.init(1e-15, 1e-6, 1e8, 5000 if not mixed else 100)
        if mixed:
            g.init_inner(1e-5, 1e-2, 1e20, 100000)
        g.build()
        if fmt != ra.CSR:
            g.convert(fmt)
        xs = g.solve(None, np.zeros(hi - lo))
        it, st, res = g.result()
        out.update(it=it, st=st, res=res, err2=float(((xs - 1.0) ** 2).sum()), xmin=float(xs.min()), xmax=float(xs.max()))
    np.savez(os.path.join(outdir, "r%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mode, rank, world, initfile, kind, outdir = sys.argv[1:7]
    if mode == "slab8":
        slab8_worker(int(rank), int(world), initfile, kind, outdir)
        sys.exit(0)
    if mode in ("amg", "amg_rccl"):
        amg_worker(int(rank), int(world), initfile, kind, outdir, rccl=(mode == "amg_rccl"))
    elif mode == "cpu":
        cpu_worker(int(rank), int(world), initfile, kind, outdir)
    elif mode == "aggregate":
        aggregate_worker(int(rank), int(world), initfile, kind, outdir)
    elif mode == "plans":
        plans_worker(int(rank), int(world), initfile, kind, outdir)
    else:
        gpu_worker(int(rank), int(world), initfile, kind, outdir, rccl=(mode == "rccl"))
