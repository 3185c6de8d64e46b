"""Krylov solvers / preconditioners: Python handles onto the compiled C++ layer
(include/rocalution/solvers.hpp, instantiated in csrc/capi_solvers.cpp).

Class and method names follow the reference (src/solvers/solver.hpp:179-444): SetOperator,
SetPreconditioner, Init, Build, Solve, Clear, GetIterationCount, GetCurrentResidual,
GetSolverStatus, GMRES.SetBasisSize, MixedPrecisionDC.Set.  The numerical work happens in the
library; nothing here computes.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (PC_GS, PC_IC, PC_ILU0, PC_SAAMG, PC_UAAMG, PC_JACOBI, PC_MCGS, PC_MCILU, PC_MCSGS, PC_NONE, PC_SGS, SOLVER_BICGSTAB,
                   SOLVER_BICGSTABL,
                   SOLVER_CG, SOLVER_CR, SOLVER_FCG, SOLVER_FGMRES, SOLVER_FIXEDPOINT, SOLVER_GMRES,
                   SOLVER_IDR, SOLVER_QMRCGSTAB)


def _lib():
    return capi.load()


TriSolverAlg_Default, TriSolverAlg_Iterative = 0, 1


class SolverDescr:
    """triangular-solve strategy of a preconditioner (solver.hpp:82-148): defaults direct, 30 sweeps, 1e-3, tol on"""

    def __init__(self):
        self.alg, self.max_iter, self.tol, self.use_tol = TriSolverAlg_Default, 30, 1e-3, True

    def SetTriSolverAlg(self, alg):
        self.alg = int(alg)

    def SetIterativeSolverMaxIteration(self, n):
        self.max_iter = int(n)

    def SetIterativeSolverTolerance(self, tol):
        self.tol = float(tol)

    def EnableIterativeSolverTolerance(self):
        self.use_tol = True

    def DisableIterativeSolverTolerance(self):
        self.use_tol = False


class _Precond:
    kind = PC_NONE

    def __init__(self):
        self.precond_format = None
        self.descr = None

    def SetSolverDescriptor(self, descr):
        self.descr = descr

    def SetPrecondMatrixFormat(self, fmt):
        self.precond_format = int(fmt)


class Jacobi(_Precond):
    kind = PC_JACOBI


class ILU(_Precond):
    kind = PC_ILU0

    def Set(self, p, level=True):
        """ILU(p) (preconditioner.cpp:420-447): fill levels on the pattern of A^(p+1), or (level=False) that whole pattern"""
        if p < 0:
            raise ValueError("ILU(p): p >= 0")
        self.params = (float(p), 1.0 if level else 0.0, 0.0)


class GS(_Precond):
    """Gauss-Seidel: LSolve on the matrix (preconditioner.cpp:206-257)"""
    kind = PC_GS


class SGS(_Precond):
    """symmetric Gauss-Seidel: LSolve, diagonal scaling, USolve (preconditioner.cpp:302-379)"""
    kind = PC_SGS


class IC(_Precond):
    """incomplete Cholesky, zero fill-in (preconditioner.cpp:826-925): ICFactorize on ExtractL, LLSolve"""
    kind = PC_IC


class UAAMG(_Precond):
    """unsmoothed-aggregation AMG as a preconditioner (unsmoothed_amg.cpp), PMIS coarsening on the device, default
    FixedPoint(2/3)+Jacobi smoothers and CG coarse solver, coarsest level <= 300 rows"""
    kind = PC_UAAMG


class SAAMG(_Precond):
    """smoothed-aggregation AMG as a preconditioner (smoothed_amg.cpp), PMIS coarsening on the device"""
    kind = PC_SAAMG


class MultiColoredSGS(_Precond):
    kind = PC_MCSGS

    def __init__(self):
        super().__init__()
        self.decomposition = True
        self.fused_sweeps = True

    def SetDecomposition(self, decomp):
        self.decomposition = bool(decomp)

    def SetFusedSweeps(self, on):
        self.fused_sweeps = bool(on)


class MultiColoredGS(MultiColoredSGS):
    """backward colour sweeps only (preconditioner_multicolored_gs.cpp:218-288)"""
    kind = PC_MCGS


class MultiColoredILU(MultiColoredSGS):
    """ILU(0,1) of the multi-coloured matrix (preconditioner_multicolored_ilu.cpp); p > 0 is not provided"""
    kind = PC_MCILU

    def Set(self, p, q=None, level=True):
        if p != 0 or (q is not None and q != 1):
            raise ValueError("only MultiColoredILU(0,1) is provided by this backend")


class _IterativeLinearSolver:
    kind = SOLVER_CG

    def __init__(self, dtype=np.float64):
        self.dtype = np.dtype(dtype)
        self._h = None
        self._op = None
        self._precond = None
        self._init = None
        self._basis = None
        self._fused = True
        self._verbose = 0

    def __del__(self):
        try:
            if self._h:
                _lib().ramd_solver_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def SetOperator(self, op):
        self._op = op

    def SetPreconditioner(self, p):
        self._precond = p

    def Init(self, abs_tol, rel_tol, div_tol, max_iter, min_iter=0):
        self._init = (float(abs_tol), float(rel_tol), float(div_tol), int(min_iter), int(max_iter))

    def InitMaxIter(self, max_iter):
        a = self._init or (1e-15, 1e-6, 1e8, 0, 1000000)
        self._init = (a[0], a[1], a[2], a[3], int(max_iter))

    def SetFused(self, on):
        self._fused = bool(on)

    def Verbose(self, v=1):
        self._verbose = int(v)

    def _create(self):
        h = C.c_void_p()
        pk = self._precond.kind if self._precond is not None else PC_NONE
        dt = capi.F64 if self.dtype == np.float64 else capi.F32
        capi.check(_lib().ramd_solver_create(self.kind, pk, dt, C.byref(h)))
        return h

    def Build(self):
        assert self._op is not None
        if self._h:
            _lib().ramd_solver_destroy(self._h)
        self._h = self._create()
        if self._init:
            capi.check(_lib().ramd_solver_init(self._h, *self._init))
        if self._basis:
            capi.check(_lib().ramd_solver_set_basis(self._h, self._basis))
        if self._precond is not None and self._precond.precond_format is not None:
            capi.check(_lib().ramd_solver_set_precond_format(self._h, self._precond.precond_format))
        if self._precond is not None and getattr(self._precond, "decomposition", True) is False:
            capi.check(_lib().ramd_solver_set_decomposition(self._h, 0))
        if self._precond is not None and getattr(self._precond, "fused_sweeps", True) is False:
            capi.check(_lib().ramd_solver_set_fused_sweeps(self._h, 0))
        if self._precond is not None and getattr(self._precond, "descr", None) is not None:
            d = self._precond.descr
            capi.check(_lib().ramd_solver_set_tri_solver(self._h, d.alg, d.max_iter, d.tol, int(d.use_tol)))
        if self._precond is not None and getattr(self._precond, "params", None) is not None:
            capi.check(_lib().ramd_solver_set_precond_params(self._h, *self._precond.params))
        capi.check(_lib().ramd_solver_set_fused(self._h, int(self._fused)))
        capi.check(_lib().ramd_solver_set_verbose(self._h, self._verbose))
        self._configure_extra()
        capi.check(_lib().ramd_solver_build(self._h, self._op._h))

    def _configure_extra(self):
        pass

    def Solve(self, rhs, x):
        capi.check(_lib().ramd_solver_solve(self._h, rhs._h, x._h))

    def PrecondApply(self, rhs, x):
        capi.check(_lib().ramd_solver_precond_apply(self._h, rhs._h, x._h))

    def ReBuildNumeric(self):
        capi.check(_lib().ramd_solver_rebuild_numeric(self._h))

    def Clear(self):
        if self._h:
            capi.check(_lib().ramd_solver_clear(self._h))

    def _result(self):
        it, st, res = C.c_int(0), C.c_int(0), C.c_double(0)
        capi.check(_lib().ramd_solver_result(self._h, C.byref(it), C.byref(st), C.byref(res)))
        return it.value, st.value, res.value

    def GetIterationCount(self):
        return self._result()[0]

    def SetTimeMark(self, iteration):
        """measurement hook: note the wall clock (device drained) when iteration `iteration` has been checked"""
        capi.check(_lib().ramd_solver_set_time_mark(self._h, int(iteration)))

    def GetSecondsSinceTimeMark(self):
        s = C.c_double(0)
        capi.check(_lib().ramd_solver_seconds_since_time_mark(self._h, C.byref(s)))
        return s.value

    def GetSolverStatus(self):
        return self._result()[1]

    def GetCurrentResidual(self):
        return self._result()[2]

    def GetNumColors(self):
        n = C.c_int(0)
        capi.check(_lib().ramd_solver_num_colors(self._h, C.byref(n)))
        return n.value

    def GetResidualHistory(self):
        n = C.c_int(0)
        capi.check(_lib().ramd_solver_history(self._h, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.float64)
        capi.check(_lib().ramd_solver_history(self._h, buf.ctypes.data_as(C.POINTER(C.c_double)), n.value,
                                              C.byref(n)))
        return buf[:n.value]


class CG(_IterativeLinearSolver):
    kind = SOLVER_CG


class GMRES(_IterativeLinearSolver):
    kind = SOLVER_GMRES

    def SetBasisSize(self, m):
        self._basis = int(m)


class BiCGStab(_IterativeLinearSolver):
    kind = SOLVER_BICGSTAB


class FCG(_IterativeLinearSolver):
    """flexible CG (src/solvers/krylov/fcg.cpp)"""
    kind = SOLVER_FCG


class CR(_IterativeLinearSolver):
    """conjugate residual (src/solvers/krylov/cr.cpp)"""
    kind = SOLVER_CR


class FGMRES(GMRES):
    """flexible GMRES (src/solvers/krylov/fgmres.cpp)"""
    kind = SOLVER_FGMRES


class BiCGStabl(_IterativeLinearSolver):
    """BiCGStab(l) (src/solvers/krylov/bicgstabl.cpp), l = 2 unless SetOrder is called"""
    kind = SOLVER_BICGSTABL

    def SetOrder(self, l):
        self._basis = int(l)


class IDR(_IterativeLinearSolver):
    """IDR(s) (src/solvers/krylov/idr.cpp); s = 4 and seed = time() unless set"""
    kind = SOLVER_IDR

    def __init__(self, dtype=np.float64):
        super().__init__(dtype)
        self._seed = None

    def SetShadowSpace(self, s):
        self._basis = int(s)

    def SetRandomSeed(self, seed):
        self._seed = int(seed)

    def _configure_extra(self):
        if self._seed is not None:
            capi.check(_lib().ramd_solver_set_seed(self._h, self._seed))


class FixedPoint(_IterativeLinearSolver):
    """x += omega M^-1 (b - A x) (src/solvers/solver.cpp:517-775); needs a preconditioner"""
    kind = SOLVER_FIXEDPOINT

    def __init__(self, dtype=np.float64):
        super().__init__(dtype)
        self._omega, self._smoother = 1.0, False

    def SetRelaxation(self, omega):
        self._omega = float(omega)

    def FlagSmoother(self):
        self._smoother = True

    def _configure_extra(self):
        capi.check(_lib().ramd_solver_set_params(self._h, self._omega, 1.0 if self._smoother else 0.0))


class QMRCGStab(_IterativeLinearSolver):
    """QMRCGStab (src/solvers/krylov/qmrcgstab.cpp)"""
    kind = SOLVER_QMRCGSTAB


class MixedPrecisionDC(_IterativeLinearSolver):
    """MixedPrecisionDC<fp64, fp32>: Set(inner) takes a float32 CG/GMRES/BiCGStab object whose
    preconditioner and Init() settings are applied to the inner solver."""

    def __init__(self):
        super().__init__(np.float64)
        self._inner = None

    def Set(self, inner):
        self._inner = inner

    def _create(self):
        h = C.c_void_p()
        pk = self._inner._precond.kind if self._inner._precond is not None else PC_NONE
        capi.check(_lib().ramd_solver_create_mixed(self._inner.kind, pk, C.byref(h)))
        return h

    def _configure_extra(self):
        if self._inner._init:
            a = self._inner._init
            capi.check(_lib().ramd_solver_init_inner(self._h, a[0], a[1], a[2], a[4]))
        if self._inner._basis:
            capi.check(_lib().ramd_solver_set_basis(self._h, self._inner._basis))
        capi.check(_lib().ramd_solver_set_fused(self._h, int(self._inner._fused)))
