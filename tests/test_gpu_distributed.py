"""GPU test of the multi-rank path on ONE device: 2 processes share the GPU and run the real
GlobalMatrix / GlobalVector / Solver<Global...> code of the library; the halo and the scalar sums
travel through the host-staged callback transport (RCCL needs one device per rank, which the 8-GPU
bench provides).  Reference: P-way result == 1-way result."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def ra():
    import rocalution_amd as ra
    ra.init_rocalution()
    return ra


@pytest.fixture(scope="module")
def S():
    from rocalution_amd import solvers
    return solvers


def _check_two_ranks(kind, oracle, mode, world=2, env=None):
    from test_cpu_host import _spawn
    import _dist_worker as W
    from rocalution_amd import generators as gen
    if kind == "poisson_slab":
        rp, ci, va = gen.poisson7(12)
    elif kind == "lap27_slab":  # the device generator of a rank's planes (ramd_mat_gen_laplace27_slab) against the whole operator
        rp, ci, va = gen.laplace27(12)
    else:
        rp, ci, va = W._matrix(kind)
        if kind == "random":
            rp, ci, va = W._symmetrize_pattern(rp, ci, va)
    n = len(rp) - 1
    x = np.random.default_rng(5).uniform(-1, 1, n)
    yref = oracle.csr_apply(rp, ci, va, x)
    b = oracle.csr_apply(rp, ci, va, np.ones(n))
    ref = oracle.solve(rp, ci, va, b, solver=oracle.CG, precond=oracle.PC_JACOBI, max_iter=500)
    res = _spawn(mode, kind, world=world, timeout=600, env=env)
    y = np.concatenate([r["y"] for r in res])
    assert np.array_equal(y, yref) or np.allclose(y, yref, rtol=1e-13, atol=1e-13)
    y_ell = np.concatenate([r["y_ell"] for r in res])
    assert np.allclose(y_ell, yref, rtol=1e-13, atol=1e-13)
    xs = np.concatenate([r["xs"] for r in res])
    assert abs(int(res[0]["it"]) - ref["iters"]) <= 1 and int(res[0]["st"]) == ref["status"]
    assert np.linalg.norm(xs - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-8
    # BiCGStab + BlockJacobi(MC-SGS), interior ELL: block-Jacobi over ranks changes the iteration count
    # with P (reference behaviour) -- check convergence to the same solution
    xs2 = np.concatenate([r["xs2"] for r in res])
    assert int(res[0]["st2"]) == 2
    assert np.linalg.norm(xs2 - 1.0) / np.sqrt(n) < 1e-4
    # BlockJacobi around SA-AMG / IC / UA-AMG (CG) or ILU(0) / MC-SGS (BiCGStab) on the interior block: converged to the
    # same solution, and in fewer iterations than the Jacobi run above
    its4 = res[0]["its4"]
    for k in range(len(its4)):
        xs4 = np.concatenate([r["xs4_%d" % k] for r in res])
        assert int(its4[k][1]) == 2, (k, its4)
        assert np.linalg.norm(xs4 - 1.0) / np.sqrt(n) < 1e-6
        assert int(its4[k][0]) < int(res[0]["it"]) * (2 if kind == "random" else 1) + 2, (k, its4, res[0]["it"])
    # P-way BlockJacobi parity (SURVEY.md 8e rule 2): iteration count, status and solution of the 2-rank run against the
    # CPU restatement of the SAME 2-way block-Jacobi algorithm (oracle nblocks=2: the preconditioner is built from the
    # rank's diagonal block only, preconditioner_blockjacobi.cpp:80-141) -- BiCGStab+BlockJacobi(MC-SGS) with ELL and HYB
    # interiors (clients/samples/bicgstab_mpi.cpp:104-116), GMRES(30)+BlockJacobi(ILU(0))
    its5 = res[0]["its5"]
    for k, (osolver, opc, slack) in enumerate(((oracle.BICGSTAB, oracle.PC_MCSGS, 2), (oracle.BICGSTAB, oracle.PC_MCSGS, 2),
                                                (oracle.GMRES, oracle.PC_ILU0, 2))):
        refb = oracle.solve(rp, ci, va, b, solver=osolver, precond=opc, max_iter=500, nblocks=world)
        xs5 = np.concatenate([r["xs5_%d" % k] for r in res])
        assert int(its5[k][1]) == refb["status"], (k, its5[k], refb["status"])
        assert abs(int(its5[k][0]) - refb["iters"]) <= slack, (k, its5[k], refb["iters"])
        d = np.linalg.norm(xs5 - refb["x"]) / np.linalg.norm(refb["x"])
        # equal iteration counts: the same iterate to round-off amplification; else both within the stopping tolerance
        assert d < (1e-8 if int(its5[k][0]) == refb["iters"] and osolver != oracle.BICGSTAB else 2e-5), (k, d)
        # ... and the P-way count differs from the 1-way one where the block structure matters (so the check is not vacuous)
    ref1 = oracle.solve(rp, ci, va, b, solver=oracle.GMRES, precond=oracle.PC_ILU0, max_iter=500)
    assert oracle.solve(rp, ci, va, b, solver=oracle.GMRES, precond=oracle.PC_ILU0, max_iter=500, nblocks=world)["iters"] >= ref1["iters"]
    # mixed precision on Global objects (no reference counterpart, SURVEY.md headline 6): pinned by the
    # 1-process MixedPrecisionDC oracle -- same outer iteration count +-1, same solution
    refm = oracle.solve_mixed(rp, ci, va, b, outer={}, inner=dict(solver=oracle.CG, precond=oracle.PC_JACOBI,
                                                                   abs_tol=1e-5, rel_tol=1e-2, div_tol=1e20,
                                                                   max_iter=100000))
    xs3 = np.concatenate([r["xs3"] for r in res])
    assert abs(int(res[0]["it3"]) - refm["iters"]) <= 1 and int(res[0]["st3"]) == refm["status"]
    assert np.linalg.norm(xs3 - refm["x"]) / np.linalg.norm(refm["x"]) < 1e-5


@pytest.mark.parametrize("kind", ["poisson_slab", "gr3030", "random", "lap27", "lap27_slab"])
def test_two_ranks_one_gpu(kind, oracle):
    _check_two_ranks(kind, oracle, "gpu")


@pytest.mark.parametrize("kind", ["random", "poisson_slab"])
def test_four_ranks_allgather_halo(kind, oracle):
    """The halo exchange as ONE all-gather of equally padded boundary buffers (the form a rank with more than four
    neighbours selects; forced here: RAMD_COMM_HALO=allgather) on a 4-way row split of a general sparse matrix -- every
    rank has up to three peers with pieces of different lengths, the random one's pattern couples all of them.  Through
    the callback transport (4 processes on one GPU): plan table exchange, padded gather, index pick; every check of the
    2-rank test (SpMV, CG+Jacobi, BlockJacobi legs against the oracle's nblocks=4 mode, mixed precision).
    Replaces the per-neighbour MPI_Isend/Irecv of src/base/parallel_manager.cpp:726-782."""
    _check_two_ranks(kind, oracle, "gpu", world=4, env={"RAMD_COMM_HALO": "allgather"})


@pytest.mark.parametrize("form", ["allgather", "sendrecv"])
def test_two_operators_whose_plans_coincide_on_one_rank_only(form):
    """Halo plans are identified by the number of the collective call that announced them (ramd_comm_halo_select), not by a
    rank's own peers and offsets: here rank 0 has NO neighbours in either of two operators on one communicator -- the same
    empty plan -- while the other three ranks exchange pieces of different lengths for the two.  With plans cached per
    rank-local signature, rank 0 would skip the second plan's table exchange and enter the all-gather with the first plan's
    padded length: a mismatched collective (hang or wrong halo).  Products with both operators in alternation and a CG
    solve with each, 4 ranks on one GPU, both forms of the exchange."""
    import scipy.sparse as sp
    from test_cpu_host import _spawn
    import _dist_worker as W
    world = 4
    res = _spawn("plans", "x", world=world, timeout=600, env={"RAMD_COMM_HALO": form})
    assert list(res[0]["npeers"]) == [0, 0] and all(min(r["npeers"]) >= 1 for r in res[1:])
    mats = W.plans_matrices(world)
    n = len(mats[0][0]) - 1
    rng = np.random.default_rng(9)
    for rep in range(3):
        x = rng.uniform(-1, 1, n)
        for k, (rp, ci, va) in enumerate(mats):
            A = sp.csr_matrix((va, ci, rp), shape=(n, n))
            y = np.concatenate([r["y%d_%d" % (k, rep)] for r in res])
            assert np.allclose(y, A @ x, rtol=1e-13, atol=1e-13), (k, rep)
    for k, (rp, ci, va) in enumerate(mats):
        A = sp.csr_matrix((va, ci, rp), shape=(n, n))
        xs = np.concatenate([r["xs%d" % k] for r in res])
        assert np.linalg.norm(xs - 1.0) / np.sqrt(n) < 1e-6  # (rhs = A * 1)


@pytest.mark.parametrize("kind", ["poisson_slab", "gr3030", "random"])
def test_two_rccl_ranks_on_two_gpus(kind, oracle):
    """The same checks with ONE GPU PER RANK and the RCCL data plane (grouped ncclSend/ncclRecv halo on the ghost stream
    overlapped with the interior SpMV, ncclAllReduce on the device scalar record): CG+Jacobi on a slab / general row split,
    BiCGStab+BlockJacobi(MC-SGS) with ELL and HYB interiors and GMRES(30)+BlockJacobi(ILU(0)) against the oracle's
    nblocks=2 mode, MixedPrecisionDC on Global objects (global_matrix.cpp:948-1008 is the choreography replaced).  Runs
    wherever at least two devices are visible (any multi-GPU driver box), skips on a 1-GPU box."""
    import rocalution_amd as ra
    if ra.device_count() < 2:
        pytest.skip("needs two GPUs (one RCCL rank per device)")
    _check_two_ranks(kind, oracle, "rccl")


def test_two_rccl_ranks_global_amg_on_two_gpus():
    """... and the aggregation AMG on the GlobalMatrix over RCCL (prolongation rows and coarse halos through grouped
    ncclSend/ncclRecv): the Galerkin identity on every level, convergence to the solution.  Skips on a 1-GPU box."""
    import rocalution_amd as ra
    if ra.device_count() < 2:
        pytest.skip("needs two GPUs (one RCCL rank per device)")
    from test_cpu_host import _spawn
    res = _spawn("amg_rccl", "poisson_slab", world=2, timeout=900)
    n = 24 ** 3
    for tag in ("ua", "sa"):
        x = np.concatenate([r["x_" + tag] for r in res])
        assert int(res[0]["res_" + tag][1]) == 2
        assert np.linalg.norm(x - 1.0) / np.sqrt(n) < 1e-6
        for r in res:
            assert r["info_" + tag][2] < 1e-12


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rccl_halo_and_allreduce_on_a_size_one_communicator(dtype):
    """RCCL refuses two ranks on one device, so the RCCL data plane is exercised with ONE rank: the halo
    exchange sends to / receives from itself (grouped ncclSend/ncclRecv on the ghost stream, ordered by
    events against the compute stream) and the scalar all-reduce runs on the device record."""
    import ctypes as C
    import subprocess
    code = r'''
import ctypes as C, os, sys
import numpy as np
os.environ["RAMD_COMM_FORCE_COLLECTIVES"] = "1"
sys.path.insert(0, %r)
import rocalution_amd as ra
from rocalution_amd import capi
ra.init_rocalution()
lib = capi.load()
dtype = np.%s
uid = C.create_string_buffer(128)
capi.check(lib.ramd_comm_unique_id(uid))
comm = C.c_void_p()
capi.check(lib.ramd_comm_init_rccl(0, 1, uid, C.byref(comm)))
n = 300000
src = np.random.default_rng(1).uniform(-1, 1, n).astype(dtype)
send = ra.LocalVector(dtype, data=src); recv = ra.LocalVector(dtype); recv.Allocate("", n)
peers = (C.c_int * 2)(0, 0)              # two "neighbours", both myself: two send/recv pairs in one group
so = (C.c_int64 * 3)(0, 100000, n); ro = (C.c_int64 * 3)(0, 100000, n)
for rep in range(3):
    recv.Zeros()
    send.Scale(2.0)                      # queued on the compute stream right before the exchange
    capi.check(lib.ramd_comm_halo_begin(comm, send._h, recv._h, 2, peers, so, ro))
    capi.check(lib.ramd_comm_halo_end(comm))
    got = recv.numpy()
    exp = (src * dtype(2.0) ** (rep + 1)).astype(dtype)
    assert np.array_equal(got, exp), rep
capi.check(lib.ramd_scalars_set(3, 1.25)); capi.check(lib.ramd_scalars_set(4, -7.5))
capi.check(lib.ramd_comm_allreduce_scalars(comm, 3, 2))
out = (C.c_double * 2)()
capi.check(lib.ramd_scalars_fetch(out, 3, 2))
assert (out[0], out[1]) == (1.25, -7.5)
capi.check(lib.ramd_comm_destroy(comm))
print("rccl self-exchange ok")
''' % (ROOT, "float64" if dtype == np.float64 else "float32")
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"rccl self-exchange ok" in r.stdout, r.stdout.decode()[-2000:]


_PM_RANK1 = """%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
%% ROCALUTION MPI ParallelManager output %%
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#RANK
1
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#GLOBAL_NROW
100
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#GLOBAL_NCOL
100
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#LOCAL_NROW
30
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#LOCAL_NCOL
30
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#BOUNDARY_SIZE
4
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#NUMBER_OF_RECEIVERS
2
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#NUMBER_OF_SENDERS
2
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#RECEIVERS_RANK
0
2
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#SENDERS_RANK
0
2
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#RECEIVERS_INDEX_OFFSET
0
2
4
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#SENDERS_INDEX_OFFSET
0
2
4
%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%%
#BOUNDARY_INDEX
0
1
28
29
"""


def test_parallel_manager_file_io(tmp_path):
    """ParallelManager::WriteFileASCII / ReadFileASCII (parallel_manager.cpp:441-743): head file + one "#KEY"-sectioned file
    per rank.  tests/drivers/pm_io_driver.cpp writes the pattern of a 3-rank chain, reads every file back into a fresh
    manager and reads the older #GLOBAL_SIZE / #LOCAL_SIZE dialect; here the text of a rank file is compared with the
    layout the reference writes (section order, separator lines, one value per line)."""
    import subprocess
    exe = str(tmp_path / "pm_io_driver")
    libdir = os.path.join(ROOT, "rocalution_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "drivers", "pm_io_driver.cpp"), "-o", exe, "-L" + libdir,
                           "-lrocalution_amd", "-Wl,-rpath," + libdir])
    d = tmp_path / "pm"
    d.mkdir()
    r = subprocess.run([exe, str(d)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"pm_io_driver ok" in r.stdout, r.stdout.decode()[-2000:]
    base = str(d / "pattern.pm")
    assert open(base).read().splitlines() == [base + ".rank.%d" % k for k in range(3)]
    assert open(base + ".rank.1").read() == _PM_RANK1


def test_distribute_matrix_one_rank_end_to_end(tmp_path):
    """distribute_matrix (include/rocalution/distribute.hpp) in the call sequence of the reference's cg_mpi sample, through a
    real RCCL communicator of size 1: same iteration count and residual as the LocalMatrix solve"""
    import subprocess
    exe = str(tmp_path / "distribute_driver")
    libdir = os.path.join(ROOT, "rocalution_amd")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "drivers", "distribute_driver.cpp"), "-o", exe, "-L" + libdir,
                           "-lrocalution_amd", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe, "20"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"distribute_driver ok" in r.stdout, r.stdout.decode()[-2000:]


def _amg_local_runs(ra, S, kind):
    import _dist_worker as W
    from rocalution_amd import generators as gen
    if kind == "poisson_slab":
        rp, ci, va = gen.poisson7(24)
    else:
        rp, ci, va = W.amg_matrix(kind)
    n = len(rp) - 1
    A = ra.LocalMatrix(); A.SetDataPtrCSR(rp, ci, va)
    b = ra.LocalVector(); b.Allocate("", n); A.Apply(ra.LocalVector(data=np.ones(n)), b)
    local = {}
    for tag, cls in (("ua", S.UAAMG), ("sa", S.SAAMG)):
        ls = S.CG(); ls.SetOperator(A); pc = cls(); ls.SetPreconditioner(pc)
        ls.Init(1e-15, 1e-8, 1e8, 200); ls.Build()
        x = ra.LocalVector(); x.Allocate("", n); x.Zeros()
        ls.Solve(b, x)
        local[tag] = (ls.GetIterationCount(), x.numpy().copy())
        ls.Clear()
    return n, local


# ("thin": slabs a few planes thick, 40 s of host-side Galerkin products -- a widened row, not the hot path: behind
#  RAMD_TEST_SOAK=1 since round 6, when the fixtures of the 27-point operator took the suite past its 600 s; the other two
#  operators and the larger-slabs / decoupled / isolated-rows tests of the same code stay in every run)
_AMG_KINDS = ["poisson_slab", "gr3030x"] + (["thin"] if os.environ.get("RAMD_TEST_SOAK", "0") not in ("", "0") else [])


@pytest.mark.parametrize("kind", _AMG_KINDS)
def test_aggregation_amg_on_the_global_matrix(ra, S, kind):
    """UAAMG / SAAMG with OperatorType = GlobalMatrix (global_matrix.cpp:1038-1880, :2607-3558; unsmoothed_amg.cpp /
    smoothed_amg.cpp), PMIS: the aggregates cross the rank boundaries as the reference's do, and -- the reference has no
    MPI here to compare with -- the property that pins the P-way algorithm is its independence of P: the aggregate
    numbering is the one a single rank produces, so
      * 1 rank: Global == Local, iteration for iteration (same kernels: the row block is the whole matrix);
      * 2 and 4 ranks: the SAME hierarchy as on one rank -- every level has the same global size, the same number of
        entries and the same || A_l 1 || (to rounding: the Galerkin sums are taken in another order), every level satisfies
        A_c x = R A_f P x on a random x (ghost parts, the halo plans of P and A_c, the reverse exchange of R), and CG takes
        the SAME number of iterations to the same solution."""
    from test_cpu_host import _spawn
    n, local = _amg_local_runs(ra, S, kind)
    worlds = (1, 3, 8) if kind == "thin" else (1, 2, 4)  # ("thin": see _dist_worker.amg_matrix)
    runs = {w: _spawn("amg", kind, world=w, timeout=900) for w in worlds}
    for tag in ("ua", "sa"):
        it1 = int(runs[1][0]["res_" + tag][0])
        x1 = runs[1][0]["x_" + tag]
        assert it1 == local[tag][0], (tag, it1, local[tag][0])
        assert np.max(np.abs(x1 - local[tag][1])) <= 1e-12
        lv1 = runs[1][0]["levels_" + tag]
        for w in worlds:
            res = runs[w]
            x = np.concatenate([r["x_" + tag] for r in res])
            it, st = int(res[0]["res_" + tag][0]), int(res[0]["res_" + tag][1])
            levels, coarsest, defect = res[0]["info_" + tag]
            assert st == 2, (tag, w, st)
            assert np.linalg.norm(x - 1.0) / np.sqrt(n) < 1e-6, (tag, w)
            assert levels >= 2 and coarsest < n / 4, (tag, w, levels, coarsest)
            for r in res:  # every rank reports the same (all-reduced) defect
                assert r["info_" + tag][2] < 1e-12, (tag, w, r["info_" + tag])
            lv = np.array([r["levels_" + tag] for r in res])  # (rank, level, what)
            assert lv.shape[1] == lv1.shape[0], (tag, w, lv.shape, lv1.shape)
            assert np.array_equal(lv[0, :, 0], lv1[:, 0]), (tag, w, lv[0, :, 0], lv1[:, 0])  # global rows per level
            assert np.array_equal(lv[:, :, 1].sum(axis=0), lv1[:, 1]), (tag, w, lv[:, :, 1].sum(axis=0), lv1[:, 1])
            assert np.allclose(lv[0, :, 2], lv1[:, 2], rtol=1e-11, atol=0), (tag, w, lv[0, :, 2], lv1[:, 2])
            assert it == it1, (tag, w, it, it1)
            assert np.max(np.abs(x - x1)) <= 1e-9, (tag, w)


@pytest.mark.parametrize("name,world,how", [("gr3030", 2, "even"), ("gr3030", 5, "uneven"), ("poisson8", 3, "even"),
                                            ("poisson8", 7, "uneven"), ("lap2d7", 4, "uneven")])
def test_distributed_pmis_aggregation_vs_oracle_and_golden(name, world, how):
    """ramd_mat_amg_pmis_aggregate_global / ramd_mat_amg_prolong_global called through the C ABI on `world` row blocks
    (global_matrix.cpp:2647-3121, :3123-3558):
      * node numbers, aggregates and root nodes over every rank's extended block (own rows, then ghost nodes) are those of
        the oracle's P-way serial mode (oracle/pmis_pway.py) -- and with it those of the genuine library's single-process
        run (tests/golden), whatever the number of ranks and wherever the block boundaries fall: bit-exact;
      * the strong-connection flags are the golden ones, entry for entry;
      * the rows of the unsmoothed prolongation put together are the golden operator bit for bit; those of the smoothed
        one agree to rounding (a row next to a block boundary adds its ghost entries after the interior ones)."""
    from conftest import load_golden
    from test_cpu_host import _spawn
    from test_oracle_golden import _pmis_oracle
    O = _pmis_oracle()
    g = load_golden(name)
    rp, ci, va = g["rowptr"], g["col"], g["val"]
    n = len(rp) - 1
    res = _spawn("aggregate", name + ":" + how, world=world, timeout=600)
    off = [int(r["lo"]) for r in res] + [n]
    oagg, oroots = O.pmis_pway(rp, ci, va, 0.01, off)
    assert np.array_equal(oagg, g["amg_agg"]) and np.array_equal(oroots, g["amg_roots"])
    nagg = int(g["amg_agg"].max()) + 1
    firsts = 0
    for r in res:
        lo, hi = int(r["lo"]), int(r["hi"])
        ext = np.concatenate([np.arange(lo, hi), r["recv_global"]]).astype(np.int64)  # global number of every node
        assert np.array_equal(r["numbers"], ext)
        assert np.array_equal(r["agg"], oagg[ext]), (lo, hi)
        assert np.array_equal(r["roots"], oroots[ext]), (lo, hi)
        assert int(r["total"]) == nagg and int(r["first"]) == firsts
        firsts += int(r["mine"])
        # strong connections: the block's entries against the golden flags of the same (row, column)
        gconn = {}
        for i in range(lo, hi):
            for j in range(rp[i], rp[i + 1]):
                gconn[(i, int(ci[j]))] = int(g["amg_conn"][j])
        brp, bci = r["blk_rp"], r["blk_ci"]
        for i in range(hi - lo):
            for j in range(brp[i], brp[i + 1]):
                assert int(r["conn"][j]) == gconn[(lo + i, int(ext[bci[j]]))]
    assert firsts == nagg
    for tag, key, exact in (("pu", "amg_P", True), ("ps", "amg_Ps", False)):
        prp = np.concatenate([[0]] + [r[tag + "_rp"][1:] + sum(len(q[tag + "_ci"]) for q in res[:k])
                                      for k, r in enumerate(res)])
        pci = np.concatenate([r[tag + "_ci"] for r in res])
        pva = np.concatenate([r[tag + "_va"] for r in res])
        assert all(int(r[tag + "_ncol"]) == nagg for r in res)
        assert np.array_equal(prp, g[key + "_rowptr"]) and np.array_equal(pci, g[key + "_col"])
        if exact:
            assert np.array_equal(pva, g[key + "_val"])
        else:
            assert np.allclose(pva, g[key + "_val"], rtol=1e-14, atol=1e-16)


# (six ranks with uneven pieces: 15-40 s of process start-up on a shared GPU -- behind RAMD_TEST_SOAK=1 like the thin-slab AMG run;
#  the four-rank run covers the same code in every run)
_PMIS_WORLDS = [(4, "even")] + ([(6, "uneven")] if os.environ.get("RAMD_TEST_SOAK", "0") not in ("", "0") else [])


@pytest.mark.parametrize("world,how", _PMIS_WORLDS)
def test_distributed_pmis_aggregation_isolated_rows_and_a_rank_without_neighbours(world, how):
    """... on a matrix without goldens, against the oracle alone: a component that fills one rank's block exactly (that rank
    exchanges with nobody while the others do), rows none of whose couplings is strong (aggregate -2, also as boundary rows
    and as ghost nodes), an irregular pattern.  Aggregates and roots bit-exact against oracle/pmis_pway.py in its P-way and
    its single-process form; the unsmoothed prolongation has its one entry per aggregated row at the aggregate's number."""
    import _dist_worker as W
    from test_cpu_host import _spawn
    from test_oracle_golden import _pmis_oracle
    O = _pmis_oracle()
    rp, ci, va = W.amg_matrix("iso")
    n = len(rp) - 1
    res = _spawn("aggregate", "iso:" + how, world=world, timeout=600)
    off = [int(r["lo"]) for r in res] + [n]
    oagg, oroots = O.pmis_pway(rp, ci, va, 0.01, off)
    _, sagg, sroots = O.pmis_single(rp, ci, va, 0.01)
    assert np.array_equal(oagg, sagg) and np.array_equal(oroots, sroots)
    assert (oagg == -2).sum() >= n // 9 and oagg.max() > 10
    if how == "even":
        assert int(res[0]["nrecv"]) == 0 and all(int(r["nrecv"]) > 0 for r in res[1:])
    nagg = int(oagg.max()) + 1
    for r in res:
        lo, hi = int(r["lo"]), int(r["hi"])
        ext = np.concatenate([np.arange(lo, hi), r["recv_global"]]).astype(np.int64)
        assert np.array_equal(r["agg"], oagg[ext]) and np.array_equal(r["roots"], oroots[ext]), (lo, hi)
        assert int(r["total"]) == nagg and int(r["pu_ncol"]) == nagg
        own = oagg[lo:hi]
        assert np.array_equal(np.diff(r["pu_rp"]), (own >= 0).astype(np.int32))
        assert np.array_equal(r["pu_ci"], own[own >= 0]) and np.all(r["pu_va"] == 1.0)


def test_aggregation_amg_on_the_global_matrix_larger_slabs():
    """... and at a size where the hierarchy is five levels deep and ranks end up with a handful of coarse rows or none
    (z-slabs of the 96^3 Poisson operator, 884 736 rows, 1 against 4 ranks): the same rows, entries and || A_l 1 || on
    every level, the same CG iteration counts (UA-AMG 43, SA-AMG 12 when this was written -- asserted equal, not to those
    numbers), the Galerkin identity on every level."""
    from test_cpu_host import _spawn
    runs = {w: _spawn("amg", "poisson_slab96", world=w, timeout=900) for w in (1, 4)}
    for tag in ("ua", "sa"):
        lv1 = runs[1][0]["levels_" + tag]
        lv = np.array([r["levels_" + tag] for r in runs[4]])
        assert lv1.shape[0] >= 4, (tag, lv1)
        assert np.array_equal(lv[0, :, 0], lv1[:, 0]) and np.array_equal(lv[:, :, 1].sum(axis=0), lv1[:, 1]), (tag, lv, lv1)
        assert np.allclose(lv[0, :, 2], lv1[:, 2], rtol=1e-11, atol=0), (tag, lv[0, :, 2], lv1[:, 2])
        it1, it4 = int(runs[1][0]["res_" + tag][0]), int(runs[4][0]["res_" + tag][0])
        assert it1 == it4 and int(runs[4][0]["res_" + tag][1]) == 2, (tag, it1, it4)
        for r in runs[4]:
            assert r["info_" + tag][2] < 1e-12, (tag, r["info_" + tag])
        x1 = runs[1][0]["x_" + tag]
        x4 = np.concatenate([r["x_" + tag] for r in runs[4]])
        assert np.max(np.abs(x4 - x1)) <= 1e-8, tag


def test_aggregation_amg_on_the_global_matrix_all_gather_halo(ra, S):
    """... with the all-gather form of the halo exchange forced (RAMD_COMM_HALO=allgather: what ranks with many neighbours
    select): the aggregation's 4-byte payloads (states, hashes, aggregate numbers: bit patterns, some of them NaNs when read
    as floats) travel through the plan's padded buffers, while the reverse exchange of the restriction and the setup messages
    stay exchanges in pairs -- same hierarchy, same iteration counts as on one rank."""
    from test_cpu_host import _spawn
    kind = "gr3030x"
    n, local = _amg_local_runs(ra, S, kind)
    runs = {w: _spawn("amg", kind, world=w, timeout=900, env={"RAMD_COMM_HALO": "allgather"}) for w in (1, 4)}
    for tag in ("ua", "sa"):
        it1 = int(runs[1][0]["res_" + tag][0])
        assert it1 == local[tag][0], (tag, it1, local[tag][0])
        res = runs[4]
        lv1 = runs[1][0]["levels_" + tag]
        lv = np.array([r["levels_" + tag] for r in res])
        assert np.array_equal(lv[0, :, 0], lv1[:, 0]) and np.array_equal(lv[:, :, 1].sum(axis=0), lv1[:, 1]), (tag, lv, lv1)
        assert np.allclose(lv[0, :, 2], lv1[:, 2], rtol=1e-11, atol=0)
        assert int(res[0]["res_" + tag][0]) == it1 and int(res[0]["res_" + tag][1]) == 2
        x = np.concatenate([r["x_" + tag] for r in res])
        assert np.max(np.abs(x - runs[1][0]["x_" + tag])) <= 1e-9
        for r in res:
            assert r["info_" + tag][2] < 1e-12


def test_aggregation_amg_on_the_global_matrix_decoupled(ra, S):
    """... and the form whose aggregates stop at the rank boundaries (RAMD_GLOBAL_AMG=decoupled; what Greedy uses on more
    than one rank): block-diagonal P and R, the Galerkin identity on every level, convergence to the solution; thinner
    blocks cost iterations (measured 10 / 11 / 16 for SA-AMG on the 120 x 120 nine-point grid over 1 / 2 / 4 ranks)."""
    from test_cpu_host import _spawn
    kind = "gr3030x"
    n, local = _amg_local_runs(ra, S, kind)
    runs = {w: _spawn("amg", kind, world=w, timeout=900, env={"RAMD_GLOBAL_AMG": "decoupled"}) for w in (1, 2, 4)}
    for tag in ("ua", "sa"):
        it1 = int(runs[1][0]["res_" + tag][0])
        assert it1 == local[tag][0], (tag, it1, local[tag][0])
        for w in (2, 4):
            res = runs[w]
            x = np.concatenate([r["x_" + tag] for r in res])
            it, st = int(res[0]["res_" + tag][0]), int(res[0]["res_" + tag][1])
            levels, coarsest, defect = res[0]["info_" + tag]
            assert st == 2, (tag, w, st)
            assert np.linalg.norm(x - 1.0) / np.sqrt(n) < 1e-6, (tag, w)
            assert levels >= 2 and coarsest < n / 4, (tag, w, levels, coarsest)
            for r in res:
                assert r["info_" + tag][2] < 1e-12, (tag, w, r["info_" + tag])
            assert it <= 2 * it1, (tag, w, it, it1)


def _bench_line(*flags, timeout=900, env=None):
    """runs bench.py as the driver does (self-spawning for --gpus N > 1) and returns its ONE JSON line"""
    import json
    import subprocess
    e = dict(os.environ)
    e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.pop("LOCAL_RANK", None)
    if env:
        e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [str(f) for f in flags], env=e, timeout=timeout,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_path_rehearsed_on_one_gpu(world):
    """The N > 1 branch of bench.py -- per-rank z-slabs, the Global solver, the HIP-event channels of the scaling leg, the
    per-rank aggregation, the max-over-ranks timing -- executed with N processes on ONE device over the host-staged
    transport (`--transport callback`), as the driver will execute it with one GPU per rank over RCCL: the JSON line is
    complete, the slabs add up to the operator, CG costs one halo exchange and two all-reduces per iteration, and the
    iterates are those of the one-rank run (same residual after the same number of iterations)."""
    N, K, W = 96, 30, 5
    one = _bench_line("--gpus", 1, "--grid", N, "--steps", K, "--warmup", W, "--no-cpu-baseline", "--no-reference-gpu",
                      "--no-extras")
    assert one["n_gpus"] == 1 and one["steps"] == K and one["value"] > 0
    out = _bench_line("--gpus", world, "--grid", N, "--steps", K, "--warmup", W, "--transport", "callback")
    assert out["n_gpus"] == world and out["steps"] == K and out["warmup"] == W
    assert out["value"] > 0 and abs(out["value"] * out["ms_per_step"] * 1e-3 - 1.0) < 1e-3
    assert out["scaling"] == "strong" and out["unit"] == "iters/s" and out["higher_is_better"] is True
    assert out["rehearsal"] is True and out["transport"].startswith("callback") and out["rccl_nranks"] == 0
    assert out["metric"] == one["metric"]  # the same metric as the 1-GPU line, on the same operator
    per = out["roofline"]["per_rank"]
    assert [r["rank"] for r in per] == list(range(world))
    assert sum(r["rows"] for r in per) == N ** 3
    assert sum(r["interior_nnz"] + r["ghost_nnz"] for r in per) == 7 * N ** 3 - 6 * N ** 2
    assert all(r["iters"] == min(K, 100) and r["spmv_avg_ms"] > 0 for r in per)
    # interior ranks talk to two neighbours, the end ranks to one: everybody exchanges once per product
    # (per iteration of the bracketed run, whose preamble -- initial residual, first direction -- adds a constant few)
    itp = per[0]["iters"]
    assert 1.0 <= out["halo_exchanges_per_iter"] <= 1.0 + 4.0 / itp
    assert 2.0 <= out["allreduces_per_iter"] <= 2.0 + 6.0 / itp  # <p,q> and {||r||^2, <r,z>} (cg.cpp:410-438: three)
    assert out["roofline"]["peak"] == 8000.0 * world and 0 < out["roofline"]["frac"] < 1
    assert out["halo_ms"] > 0 and out["halo_overlap_frac"] is not None
    # the same Krylov iterates: residual after W + K iterations agrees with the one-rank run (dots summed in another order)
    assert out["final_residual"] == pytest.approx(one["final_residual"], rel=1e-6)


@pytest.mark.parametrize("flags", [("--solver", "bicgstab", "--precond", "mcsgs", "--format", "ell"),
                                   ("--solver", "mixed"),
                                   ("--solver", "gmres", "--precond", "ilu0")])
def test_bench_other_configs_multi_rank_rehearsal(flags):
    """configs 4 / 5 and north_star's second solver through the same N > 1 branch (2 ranks on one device)"""
    out = _bench_line("--gpus", 2, "--grid", 64, "--steps", 12, "--warmup", 3, "--transport", "callback", *flags)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["rehearsal"] is True
    assert sum(r["rows"] for r in out["roofline"]["per_rank"]) == 64 ** 3
    assert np.isfinite(out["final_residual"])


def test_bench_under_torch_distributed_run_rehearsed_on_one_gpu():
    """the launch form the driver uses for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher, rank 0
    prints the one JSON line -- with the host-staged transport on one device; a launcher world size that differs from
    --gpus is refused"""
    import json
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    p = subprocess.run(base + ["--gpus", "2", "--grid", "64", "--steps", "12", "--warmup", "3", "--transport", "callback"], env=e,
                       timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["value"] > 0 and out["rehearsal"] is True
    assert sum(r["rows"] for r in out["roofline"]["per_rank"]) == 64 ** 3
    p = subprocess.run(base + ["--gpus", "4", "--grid", "64", "--steps", "12", "--warmup", "3", "--transport", "callback"], env=e,
                       timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "refusing" in p.stderr
