"""BASELINE.json configs 4 and 5 at FULL size (3-D Poisson 512^3) through the GlobalMatrix / GlobalVector path on one GPU:
a size-1 RCCL communicator with the collectives forced on (RAMD_COMM_FORCE_COLLECTIVES=1), i.e. the code path of the
multi-GPU run with every ncclAllReduce really issued.  The oracle cannot run 134M rows in test time; checked instead:
  * C4: BiCGStab + BlockJacobi(MultiColoredSGS), interior converted to ELL and to HYB after Build() (the reference's
    test order, clients/include/testing_cg.hpp:151-155; solver set-up of clients/samples/bicgstab_mpi.cpp:104-116):
    converges to x = 1, iteration count in the range the 32^3 / 48^3 oracle runs extrapolate to (25 / 36 iterations:
    ~0.75 N), and the SAME iteration count, final residual and solution as the LocalMatrix path at one rank
    (SURVEY.md 8e rule 3: P = 1 Global == Local)
  * C5: MixedPrecisionDC (fp64 defect correction around fp32 CG + Jacobi, inner Init(1e-5, 1e-2, 1e20, 100000) as
    clients/samples/mixed-precision.cpp:85) converges to x = 1 in a handful of outer iterations.
Runs in a subprocess: the force switch is read once per process.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PRELUDE = r'''
import ctypes as C, os, sys
import numpy as np
os.environ["RAMD_COMM_FORCE_COLLECTIVES"] = "1"
sys.path.insert(0, %r)
import rocalution_amd as ra
from rocalution_amd import capi, solvers as S
ra.init_rocalution()
lib = capi.load()
N = 512
n = N ** 3
uid = C.create_string_buffer(128)
capi.check(lib.ramd_comm_unique_id(uid))
comm = C.c_void_p()
capi.check(lib.ramd_comm_init_rccl(0, 1, uid, C.byref(comm)))
nr = C.c_int(0); capi.check(lib.ramd_comm_rccl_count(comm, C.byref(nr))); assert nr.value == 1

def gsolve(g, fmt):
    capi.check(lib.ramd_gsolver_setup_poisson(g, N, 0, N))
    capi.check(lib.ramd_gsolver_build(g))
    if fmt != ra.CSR:
        capi.check(lib.ramd_gsolver_convert(g, fmt))
    capi.check(lib.ramd_prof_enable(4, 1))  # count the all-reduces of the run
    x = np.zeros(n)
    capi.check(lib.ramd_gsolver_solve(g, None, x.ctypes.data_as(C.c_void_p)))  # rhs = A*1, x0 = 0
    it, st, rs = C.c_int(0), C.c_int(0), C.c_double(0)
    capi.check(lib.ramd_gsolver_result(g, C.byref(it), C.byref(st), C.byref(rs)))
    cnt = C.c_int64(0); capi.check(lib.ramd_prof_count(4, C.byref(cnt)))
    return x, it.value, st.value, rs.value, cnt.value
''' % ROOT


def _run(body, timeout=1500):
    r = subprocess.run([sys.executable, "-c", _PRELUDE + body], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode()
    assert r.returncode == 0 and "FULL-SIZE OK" in out, out[-3000:]
    return out


@pytest.mark.parametrize("fmt", ["ELL", "HYB"])
def test_config4_bicgstab_blockjacobi_mcsgs_512_global_equals_local(fmt):
    _run(r'''
fmt = ra.%s
g = C.c_void_p()
capi.check(lib.ramd_gsolver_create(comm, capi.SOLVER_BICGSTAB, capi.PC_MCSGS, C.byref(g)))
capi.check(lib.ramd_gsolver_init(g, 1e-15, 1e-6, 1e8, 0, 5000))
xg, itg, stg, rsg, nred = gsolve(g, fmt)
assert stg == 2 and 250 <= itg <= 700, (itg, stg)      # 32^3: 25, 48^3: 36 iterations -> ~0.75 N
assert nred >= 3 * itg, (nred, itg)                     # the scalar all-reduces really went through RCCL
# (a run stopped at a relative residual of 1e-6 on an operator of condition ~1e5; BiCGStab's path depends on the rounding
#  of its dots: 349 iterations / rms 7.1e-5 and 367 / 1.2e-4 with two orders of the fused dot's partial sums, r03cm)
assert np.sqrt(np.mean((xg - 1.0) ** 2)) < 5e-4, np.sqrt(np.mean((xg - 1.0) ** 2))
capi.check(lib.ramd_gsolver_destroy(g))
# the LocalMatrix path on the same operator: same iterates
A = ra.LocalMatrix(); A.GenPoisson7(N)
ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
rhs = ra.LocalVector(); rhs.Allocate("", n); A.Apply(ones, rhs)
ls = S.BiCGStab(); ls.SetOperator(A); ls.SetPreconditioner(S.MultiColoredSGS()); ls.Init(1e-15, 1e-6, 1e8, 5000); ls.Build()
assert A.ConvertTo(fmt) == fmt
x = ra.LocalVector(); x.Allocate("", n)
ls.Solve(rhs, x)
assert ls.GetSolverStatus() == 2
assert ls.GetIterationCount() == itg, (ls.GetIterationCount(), itg)
assert abs(ls.GetCurrentResidual() - rsg) <= 1e-9 * rsg, (ls.GetCurrentResidual(), rsg)
assert np.max(np.abs(x.numpy() - xg)) <= 1e-9
print("FULL-SIZE OK", itg, rsg, nred)
''' % fmt)


def test_config5_mixed_precision_512_global():
    _run(r'''
g = C.c_void_p()
capi.check(lib.ramd_gsolver_create_mixed(comm, capi.SOLVER_CG, capi.PC_JACOBI, C.byref(g)))
capi.check(lib.ramd_gsolver_init(g, 1e-15, 1e-6, 1e8, 0, 100))
capi.check(lib.ramd_gsolver_init_inner(g, 1e-5, 1e-2, 1e20, 100000))
xg, itg, stg, rsg, nred = gsolve(g, ra.CSR)
assert stg == 2 and 2 <= itg <= 8, (itg, stg)           # 32^3 .. 128^3 oracle runs: 3-4 outer iterations
assert nred > 100                                        # inner fp32 CG: two all-reduces per iteration
assert np.sqrt(np.mean((xg - 1.0) ** 2)) < 1e-4
capi.check(lib.ramd_gsolver_destroy(g))
# the LocalMatrix MixedPrecisionDC on the same operator: same outer iteration count
A = ra.LocalMatrix(); A.GenPoisson7(N)
ones = ra.LocalVector(); ones.Allocate("", n); ones.Ones()
rhs = ra.LocalVector(); rhs.Allocate("", n); A.Apply(ones, rhs)
inner = S.CG(np.float32); inner.SetPreconditioner(S.Jacobi()); inner.Init(1e-5, 1e-2, 1e20, 100000)
mp = S.MixedPrecisionDC(); mp.SetOperator(A); mp.Set(inner); mp.Init(1e-15, 1e-6, 1e8, 100); mp.Build()
x = ra.LocalVector(); x.Allocate("", n)
mp.Solve(rhs, x)
assert mp.GetSolverStatus() == 2 and abs(mp.GetIterationCount() - itg) <= 1, (mp.GetIterationCount(), itg)
assert np.sqrt(np.mean((x.numpy() - 1.0) ** 2)) < 1e-4
print("FULL-SIZE OK", itg, rsg, nred)
''')


# ------------------------------------------------------------------ the TARGET decomposition: 8 ranks
# BASELINE.json configs 4 and 5 are "row-split across 8 x MI355X".  One device per lease here, so the 8 ranks share it over the
# host-staged callback transport -- everything else is what an 8-GPU node runs: eight 64-plane slabs, their halo plans
# (interior ranks two neighbours, the end ranks one), 8-block BlockJacobi, all-reduced scalars.
def _spawn8(kind, timeout=1500):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_cpu_host import _spawn
    return _spawn("slab8", kind, world=8, timeout=timeout)


def test_eight_way_global_apply_is_the_closed_form_512():
    """GlobalMatrix::Apply on 8 slabs of the 512^3 operator (global_matrix.cpp:924-1009): A*1 = 6 - #neighbours and A*x for an
    integer-valued x (every product and sum exact, so any order of the interior / ghost parts gives the same bits), in CSR and
    with the interior converted to ELL and to HYB -- bit for bit on every rank, incl. the rows next to the rank boundaries."""
    res = _spawn8("apply:512")
    assert [int(r["lo"]) for r in res] == [k * 64 * 512 * 512 for k in range(8)]
    assert all(bool(r["apply_ones_ok"]) and bool(r["apply_x_ok"]) for r in res), [(bool(r["apply_ones_ok"]), bool(r["apply_x_ok"])) for r in res]


def test_eight_ranks_config4_iteration_count_is_the_oracles_8_block_count_128(oracle):
    """BiCGStab + BlockJacobi(MC-SGS), interior in ELL, 8 ranks at 128^3 against the CPU restatement of the SAME 8-way
    block-Jacobi algorithm (oracle nblocks = 8: preconditioner_blockjacobi.cpp:80-141)."""
    from rocalution_amd import generators as gen
    N = 128
    rp, ci, va = gen.poisson7(N)
    b = oracle.csr_apply(rp, ci, va, np.ones(N ** 3))
    ref = oracle.solve(rp, ci, va, b, solver=oracle.BICGSTAB, precond=oracle.PC_MCSGS, max_iter=5000, nblocks=8)
    ref1 = oracle.solve(rp, ci, va, b, solver=oracle.BICGSTAB, precond=oracle.PC_MCSGS, max_iter=5000)
    res = _spawn8("c4:128", timeout=900)
    its = [int(r["it"]) for r in res]
    assert len(set(its)) == 1 and all(int(r["st"]) == ref["status"] == 2 for r in res), (its, ref["status"])
    assert len(set(float(r["res"]) for r in res)) == 1  # the all-reduced residual: the same number on every rank
    # (BiCGStab's path moves with the order of the partial sums of its dots -- tests/test_oracle_golden.py measures that spread
    #  on the oracle itself -- hence the slack of the P-way parity bar)
    assert abs(its[0] - ref["iters"]) <= 2, (its[0], ref["iters"], ref1["iters"])
    err = np.sqrt(sum(float(r["err2"]) for r in res) / N ** 3)
    assert err < 1e-4, err
    print("8 ranks: %d iterations, oracle nblocks=8: %d, nblocks=1: %d" % (its[0], ref["iters"], ref1["iters"]))


SOAK = os.environ.get("RAMD_TEST_SOAK", "0") not in ("", "0")
# full convergence of four solver / preconditioner pairs on the 8-way split: config 4 runs at BASELINE's 512^3 in every run; the
# others converge at 256^3 by default and at 512^3 -- 170 s more of the driver's 1 200-s limit, GMRES alone 110 s -- with
# RAMD_TEST_SOAK=1 ("soak loops behind an env switch").  The size-independent claims (one iteration count, one
# all-reduced residual on every rank, x -> 1) are the same at both sizes.
CONVERGE = [("c4", 512), ("c4hyb", 256), ("c5", 256), ("gmres", 256)] + ([("c4hyb", 512), ("c5", 512), ("gmres", 512)] if SOAK else [])


@pytest.mark.parametrize("what,N", CONVERGE)
def test_eight_ranks_converge_at_512(what, N):
    """configs 4 (ELL and HYB interior) and 5, and GMRES(30) + BlockJacobi(ILU(0)), on the 8-way split of N^3: converge to
    x = 1, every rank reporting the same iteration count, status and (all-reduced) residual"""
    res = _spawn8("%s:%d" % (what, N), timeout=2400)
    its = [int(r["it"]) for r in res]
    assert len(set(its)) == 1 and all(int(r["st"]) == 2 for r in res), (its, [int(r["st"]) for r in res])
    assert len(set(float(r["res"]) for r in res)) == 1
    rms = np.sqrt(sum(float(r["err2"]) for r in res) / N ** 3)
    assert rms < (1e-4 if what == "c5" else 1e-3), rms
    lim = {"c4": (250, 900), "c4hyb": (250, 900), "c5": (2, 8), "gmres": (300, 3000)}[what]
    if N < 512:
        lim = (lim[0] // 3, lim[1])
    assert lim[0] <= its[0] <= lim[1], its
    print(what, "8 ranks at %d^3:" % N, its[0], "iterations, residual", float(res[0]["res"]), "rms error", rms)
