"""CPU-side tests (no GPU): the C ABI surface, the C++ API layer's drop-in compile, and the host logic
of the row-block decomposition exercised with 2 gloo ranks."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "rocalution_amd.h")


def _declared():
    return set(re.findall(r"\b(ramd_[a-z0-9_]+)\s*\(", open(HDR).read())) - {"ramd_exchange_cb", "ramd_allreduce_cb"}


def test_abi_header_library_and_ctypes_table_agree():
    from rocalution_amd import build, capi
    lib = build.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    exported = set(re.findall(r" T (ramd_[a-z0-9_]+)", out))
    declared = _declared()
    assert declared == exported, (sorted(declared - exported), sorted(exported - declared))
    assert set(capi.SIGNATURES) == declared
    capi.load()  # dlopen + every prototype attached


def test_library_refuses_to_run_without_a_gpu():
    from rocalution_amd import capi
    lib = capi.load()
    import ctypes as C
    c = C.c_int(0)
    lib.ramd_device_count(C.byref(c))
    if c.value > 0:
        pytest.skip("a GPU is present")
    assert lib.ramd_init(-1) == capi.ERR_NO_DEVICE
    assert b"no host compute path" in lib.ramd_last_error()
    import rocalution_amd as ra
    with pytest.raises(ra.RamdError):
        ra.LocalVector()  # no silent CPU fallback anywhere


DRIVER = r"""
// a driver written against the reference API (cf. clients/samples/cg.cpp, gmres.cpp, bicgstab.cpp,
// mixed-precision.cpp, cg_mpi.cpp) -- must compile with a plain host compiler, no HIP headers
#include <rocalution/rocalution.hpp>
using namespace rocalution;
int main(int argc, char** argv)
{
    init_rocalution();
    info_rocalution();
    LocalVector<double> x, rhs, e;
    LocalMatrix<double> mat;
    mat.ReadFileMTX(std::string(argv[1]));
    mat.MoveToAccelerator(); x.MoveToAccelerator(); rhs.MoveToAccelerator(); e.MoveToAccelerator();
    x.Allocate("x", mat.GetN()); rhs.Allocate("rhs", mat.GetM()); e.Allocate("e", mat.GetN());
    CG<LocalMatrix<double>, LocalVector<double>, double> ls;
    Jacobi<LocalMatrix<double>, LocalVector<double>, double> p;
    e.Ones(); mat.Apply(e, &rhs); x.Zeros();
    ls.SetOperator(mat); ls.SetPreconditioner(p); ls.Build(); ls.Verbose(1);
    mat.Info();
    double tick = rocalution_time();
    ls.Solve(rhs, &x);
    double tack = rocalution_time();
    std::cout << "Solver execution:" << (tack - tick) / 1e6 << " sec" << std::endl;
    ls.Clear();
    GMRES<LocalMatrix<double>, LocalVector<double>, double> gm;
    ILU<LocalMatrix<double>, LocalVector<double>, double> ilu;
    gm.SetOperator(mat); gm.SetBasisSize(30); gm.SetPreconditioner(ilu); gm.Build(); x.Zeros(); gm.Solve(rhs, &x); gm.Clear();
    BiCGStab<LocalMatrix<double>, LocalVector<double>, double> bi;
    MultiColoredSGS<LocalMatrix<double>, LocalVector<double>, double> sgs;
    bi.SetOperator(mat); bi.SetPreconditioner(sgs); bi.Build(); mat.ConvertToELL(); x.Zeros(); bi.Solve(rhs, &x); bi.Clear();
    MixedPrecisionDC<LocalMatrix<double>, LocalVector<double>, double, LocalMatrix<float>, LocalVector<float>, float> mp;
    CG<LocalMatrix<float>, LocalVector<float>, float> cgf;
    cgf.Init(1e-5, 1e-2, 1e+20, 100000);
    mat.ConvertToCSR(); mp.SetOperator(mat); mp.Set(cgf); mp.Build(); x.Zeros(); mp.Solve(rhs, &x); mp.Clear();
    ParallelManager pm; GlobalMatrix<double> gmat(pm); GlobalVector<double> gx(pm), grhs(pm);
    CG<GlobalMatrix<double>, GlobalVector<double>, double> gls;
    BlockJacobi<GlobalMatrix<double>, GlobalVector<double>, double> bj;
    bj.Set(sgs); gls.SetOperator(gmat); gls.SetPreconditioner(bj);
    e.ScaleAdd(-1.0, x);
    std::cout << "||e - x||_2 = " << e.Norm() << std::endl;
    stop_rocalution();
    return 0;
}
"""


def test_reference_style_driver_compiles_with_plain_gxx():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "driver.cpp")
        open(src, "w").write(DRIVER)
        subprocess.check_call(["g++", "-std=c++14", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), src])


def test_partition_matches_reference_rule():
    from rocalution_amd import distributed as D
    for n, p in ((10, 3), (900, 2), (7, 8), (512 ** 3, 8)):
        off = D.partition_rows(n, p)
        sizes = np.diff(off)
        assert off[0] == 0 and off[-1] == n and sizes.max() - sizes.min() <= 1
        assert np.all(sizes[: n % p] == n // p + 1)  # the first nrow % P ranks get one more row


def test_split_rows_and_halo_plan_single_process():
    """assemble all ranks' pieces in one process and check y = A x piecewise"""
    from rocalution_amd import distributed as D, generators as gen
    rp, ci, va = gen.poisson7(6)
    n = len(rp) - 1
    import scipy.sparse as sp
    A = sp.csr_matrix((va, ci, rp), shape=(n, n))
    x = np.random.default_rng(0).uniform(-1, 1, n)
    for world in (1, 2, 3, 5):
        off = D.partition_rows(n, world)
        pieces = [D.split_rows(rp, ci, va, off, r) for r in range(world)]
        needs = [{int(p): pc["recv_global"][pc["recv_offset"][k]:pc["recv_offset"][k + 1]]
                  for k, p in enumerate(pc["recv_peers"])} for pc in pieces]
        for r, pc in enumerate(pieces):
            plan = D.build_halo_plan(pc, off, r, lambda obj: needs)
            lo, hi = pc["row_begin"], pc["row_end"]
            irp, ici, iva = pc["interior"]
            grp, gci, gva = pc["ghost"]
            recv = x[pc["recv_global"]]
            y = sp.csr_matrix((iva, ici, irp), shape=(hi - lo, hi - lo)) @ x[lo:hi]
            if len(gva):
                y = y + sp.csr_matrix((gva, gci, grp), shape=(hi - lo, len(recv))) @ recv
            assert np.allclose(y, (A @ x)[lo:hi], rtol=1e-14, atol=1e-14)
            # what I send is what my neighbours expect to receive, in their order
            for k, q in enumerate(plan["peers"]):
                mine = plan["boundary_index"][plan["send_offset"][k]:plan["send_offset"][k + 1]] + lo
                assert np.array_equal(mine, needs[int(q)][r])


def _spawn(mode, kind, world=2, timeout=300):
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), mode, str(r),
                                   str(world), initfile, kind, d]) for r in range(world)]
        for p in procs:
            assert p.wait(timeout=timeout) == 0
        return [dict(np.load(os.path.join(d, "r%d.npz" % r))) for r in range(world)]


@pytest.mark.parametrize("kind", ["poisson", "gr3030", "random"])
def test_two_rank_gloo_spmv_and_cg_match_single_rank(kind, oracle):
    """world_size-2 gloo run of the decomposition (product host logic + oracle kernels) == 1-rank oracle"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _dist_worker as W
    rp, ci, va = W._matrix(kind)
    if kind == "random":
        rp, ci, va = W._symmetrize_pattern(rp, ci, va)
    n = len(rp) - 1
    x = np.random.default_rng(5).uniform(-1, 1, n)
    yref = oracle.csr_apply(rp, ci, va, x)
    b = oracle.csr_apply(rp, ci, va, np.ones(n))
    ref = oracle.solve(rp, ci, va, b, solver=oracle.CG, precond=oracle.PC_JACOBI, max_iter=500)
    res = _spawn("cpu", kind)
    y = np.concatenate([r["y"] for r in res])
    xs = np.concatenate([r["xs"] for r in res])
    assert np.allclose(y, yref, rtol=1e-13, atol=1e-13)
    assert abs(int(res[0]["it"]) - ref["iters"]) <= 1
    assert np.linalg.norm(xs - ref["x"]) / np.linalg.norm(ref["x"]) < 1e-8
