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


@pytest.mark.parametrize("kind", ["poisson_slab", "gr3030", "random"])
def test_two_ranks_one_gpu(kind, oracle):
    from test_cpu_host import _spawn
    import _dist_worker as W
    from rocalution_amd import generators as gen
    if kind == "poisson_slab":
        rp, ci, va = gen.poisson7(12)
    else:
        rp, ci, va = W._matrix(kind)
        if kind == "random":
            rp, ci, va = W._symmetrize_pattern(rp, ci, va)
    n = len(rp) - 1
    x = np.random.default_rng(5).uniform(-1, 1, n)
    yref = oracle.csr_apply(rp, ci, va, x)
    b = oracle.csr_apply(rp, ci, va, np.ones(n))
    ref = oracle.solve(rp, ci, va, b, solver=oracle.CG, precond=oracle.PC_JACOBI, max_iter=500)
    res = _spawn("gpu", kind)
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
    # mixed precision on Global objects (no reference counterpart, SURVEY.md headline 6): pinned by the
    # 1-process MixedPrecisionDC oracle -- same outer iteration count +-1, same solution
    refm = oracle.solve_mixed(rp, ci, va, b, outer={}, inner=dict(solver=oracle.CG, precond=oracle.PC_JACOBI,
                                                                   abs_tol=1e-5, rel_tol=1e-2, div_tol=1e20,
                                                                   max_iter=100000))
    xs3 = np.concatenate([r["xs3"] for r in res])
    assert abs(int(res[0]["it3"]) - refm["iters"]) <= 1 and int(res[0]["st3"]) == refm["status"]
    assert np.linalg.norm(xs3 - refm["x"]) / np.linalg.norm(refm["x"]) < 1e-5
