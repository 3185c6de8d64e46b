"""rocalution_amd -- MI355X-native preconditioned-Krylov core behind the rocALUTION API surface.

Python mirror of the reference's front-end objects for the hot path (same method names and
argument meaning as src/base/local_vector.hpp / local_matrix.hpp): thin handles over the C ABI of
librocalution_amd.so (include/rocalution_amd.h).  There is no host compute path: every object
lives on the accelerator, and the library refuses to initialise without a GPU.

The Krylov solvers / preconditioners (rocalution_amd.solvers) are the compiled C++ layer
(include/rocalution/*.hpp) exported through the same library.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import CSR, COO, ELL, HYB, F32, F64, I32, RamdError  # noqa: F401

_NP = {F64: np.float64, F32: np.float32, I32: np.int32}
_DT = {np.dtype(np.float64): F64, np.dtype(np.float32): F32, np.dtype(np.int32): I32}


def _lib():
    return capi.load()


def init_rocalution(device=-1):
    """init_rocalution (src/base/backend_manager.cpp:110): selects the device and creates streams."""
    capi.check(_lib().ramd_init(int(device)))


def stop_rocalution():
    capi.check(_lib().ramd_stop())


def info_rocalution():
    buf = C.create_string_buffer(512)
    capi.check(_lib().ramd_info(buf, 512))
    return buf.value.decode()


def device_count():
    c = C.c_int(0)
    _lib().ramd_device_count(C.byref(c))
    return c.value


def sync():
    """_rocalution_sync()"""
    capi.check(_lib().ramd_sync())


def _offscope(name):
    """entry points outside SURVEY.md's scope exist only in a library built with -DRAMD_WITH_OFFSCOPE (capi.OPTIONAL)"""
    if not capi.has(name):
        raise NotImplementedError(name + ": out of scope (SURVEY.md section 2), not in the default build of librocalution_amd.so; "
                                  "rebuild with RAMD_EXTRA_CXXFLAGS=-DRAMD_WITH_OFFSCOPE")
    return getattr(_lib(), name)


class LocalVector:
    """LocalVector<ValueType> resident on the accelerator (src/base/local_vector.hpp)."""

    def __init__(self, dtype=np.float64, data=None):
        self.dtype = np.dtype(dtype)
        self._h = capi.vec_t()
        capi.check(_lib().ramd_vec_create(_DT[self.dtype], C.byref(self._h)))
        if data is not None:
            data = np.ascontiguousarray(data, dtype=self.dtype)
            self.Allocate("", data.size)
            self.CopyFromHostData(data)

    def __del__(self):
        try:
            if self._h:
                _lib().ramd_vec_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- allocation / data movement
    def Allocate(self, name, n):
        capi.check(_lib().ramd_vec_allocate(self._h, int(n)))

    def Clear(self):
        capi.check(_lib().ramd_vec_clear(self._h))

    def GetSize(self):
        n = C.c_int64(0)
        capi.check(_lib().ramd_vec_size(self._h, C.byref(n)))
        return n.value

    def CopyFromHostData(self, data):
        data = np.ascontiguousarray(data, dtype=self.dtype)
        assert data.size == self.GetSize()
        capi.check(_lib().ramd_vec_copy_from_host(self._h, data.ctypes.data_as(C.c_void_p)))

    def CopyToHostData(self):
        out = np.empty(self.GetSize(), dtype=self.dtype)
        capi.check(_lib().ramd_vec_copy_to_host(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    numpy = CopyToHostData

    def Zeros(self):
        capi.check(_lib().ramd_vec_zeros(self._h))

    def Ones(self):
        capi.check(_lib().ramd_vec_ones(self._h))

    def SetValues(self, val):
        capi.check(_lib().ramd_vec_set_values(self._h, float(val)))

    def CopyFrom(self, src, src_offset=None, dst_offset=None, size=None):
        if src_offset is None:
            capi.check(_lib().ramd_vec_copy_from(self._h, src._h))
        else:
            capi.check(_lib().ramd_vec_copy_from_offset(self._h, src._h, int(src_offset), int(dst_offset),
                                                        int(size)))

    def CopyFromFloat(self, src):
        capi.check(_lib().ramd_vec_copy_from_float(self._h, src._h))

    def CopyFromDouble(self, src):
        capi.check(_lib().ramd_vec_copy_from_double(self._h, src._h))

    def CopyFromPermute(self, src, perm):
        capi.check(_lib().ramd_vec_copy_from_permute(self._h, src._h, perm._h))

    def CopyFromPermuteBackward(self, src, perm):
        capi.check(_lib().ramd_vec_copy_from_permute_backward(self._h, src._h, perm._h))

    # -- BLAS-1
    def AddScale(self, x, alpha):
        capi.check(_lib().ramd_vec_add_scale(self._h, x._h, float(alpha)))

    def ScaleAdd(self, alpha, x):
        capi.check(_lib().ramd_vec_scale_add(self._h, float(alpha), x._h))

    def ScaleAddScale(self, alpha, x, beta):
        capi.check(_lib().ramd_vec_scale_add_scale(self._h, float(alpha), x._h, float(beta)))

    def ScaleAdd2(self, alpha, x, beta, y, gamma):
        capi.check(_lib().ramd_vec_scale_add2(self._h, float(alpha), x._h, float(beta), y._h, float(gamma)))

    def Scale(self, alpha):
        capi.check(_lib().ramd_vec_scale(self._h, float(alpha)))

    def Dot(self, x):
        r = C.c_double(0)
        capi.check(_lib().ramd_vec_dot(self._h, x._h, C.byref(r)))
        return r.value

    DotNonConj = Dot

    def ReadFileASCII(self, filename):
        """LocalVector file IO with the reference's formats (host_vector.cpp:415-632)"""
        capi.check(_lib().ramd_vec_read_file(self._h, str(filename).encode(), 0))

    def ReadFileBinary(self, filename):
        capi.check(_lib().ramd_vec_read_file(self._h, str(filename).encode(), 1))

    def WriteFileASCII(self, filename):
        capi.check(_lib().ramd_vec_write_file(self._h, str(filename).encode(), 0))

    def WriteFileBinary(self, filename):
        capi.check(_lib().ramd_vec_write_file(self._h, str(filename).encode(), 1))

    def Norm(self):
        r = C.c_double(0)
        capi.check(_lib().ramd_vec_norm(self._h, C.byref(r)))
        return r.value

    def Reduce(self):
        r = C.c_double(0)
        capi.check(_lib().ramd_vec_reduce(self._h, C.byref(r)))
        return r.value

    def Asum(self):
        r = C.c_double(0)
        capi.check(_lib().ramd_vec_asum(self._h, C.byref(r)))
        return r.value

    def Amax(self):
        r, i = C.c_double(0), C.c_int64(0)
        capi.check(_lib().ramd_vec_amax(self._h, C.byref(r), C.byref(i)))
        return i.value, r.value

    def PointWiseMult(self, x, y=None):
        if y is None:
            capi.check(_lib().ramd_vec_pointwise_mult(self._h, x._h))
        else:
            capi.check(_lib().ramd_vec_pointwise_mult2(self._h, x._h, y._h))

    def GetIndexValues(self, index, out):
        capi.check(_lib().ramd_vec_get_index_values(self._h, index._h, out._h))


class LocalMatrix:
    """LocalMatrix<ValueType> resident on the accelerator (src/base/local_matrix.hpp)."""

    def __init__(self, dtype=np.float64, _handle=None):
        self.dtype = np.dtype(dtype)
        if _handle is not None:
            self._h = _handle
        else:
            self._h = capi.mat_t()
            capi.check(_lib().ramd_mat_create(_DT[self.dtype], C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                _lib().ramd_mat_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _info(self):
        nr, nc, fmt, dt = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        nnz = C.c_int64(0)
        capi.check(_lib().ramd_mat_info(self._h, C.byref(nr), C.byref(nc), C.byref(nnz), C.byref(fmt),
                                        C.byref(dt)))
        return nr.value, nc.value, nnz.value, fmt.value

    def GetM(self):
        return self._info()[0]

    def GetN(self):
        return self._info()[1]

    def GetNnz(self):
        return self._info()[2]

    def GetFormat(self):
        return self._info()[3]

    def Clear(self):
        capi.check(_lib().ramd_mat_clear(self._h))

    def SetDataPtrCSR(self, row_offset, col, val, name="", nnz=None, nrow=None, ncol=None):
        """(copies; the reference steals the pointers -- ownership is moot across ctypes)"""
        rp = np.ascontiguousarray(row_offset, dtype=np.int32)
        ci = np.ascontiguousarray(col, dtype=np.int32)
        va = np.ascontiguousarray(val, dtype=self.dtype)
        nrow = len(rp) - 1 if nrow is None else nrow
        ncol = nrow if ncol is None else ncol
        capi.check(_lib().ramd_mat_set_csr_from_host(self._h, int(nrow), int(ncol), int(len(va)),
                                                     rp.ctypes.data_as(C.c_void_p),
                                                     ci.ctypes.data_as(C.c_void_p),
                                                     va.ctypes.data_as(C.c_void_p)))

    CopyFromCSR = SetDataPtrCSR

    def SetDataPtrCOO(self, row, col, val, name="", nnz=None, nrow=None, ncol=None):
        """LocalMatrix::SetDataPtrCOO: entries go to their rows by a STABLE sort (a row's products are still
        added in storage order, as in the reference's serial COO loop); the object is a COO matrix afterwards"""
        row = np.ascontiguousarray(row, dtype=np.int64)
        order = np.argsort(row, kind="stable")
        nrow = int(row.max()) + 1 if nrow is None else int(nrow)
        rp = np.zeros(nrow + 1, dtype=np.int64)
        np.cumsum(np.bincount(row, minlength=nrow), out=rp[1:])
        self.SetDataPtrCSR(rp, np.asarray(col)[order], np.asarray(val)[order], nrow=nrow, ncol=ncol)
        if len(row):
            self.ConvertTo(COO)

    def CopyToCSR(self):
        nr, nc, nnz, fmt = self._info()
        rp = np.empty(nr + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        va = np.empty(nnz, dtype=self.dtype)
        capi.check(_lib().ramd_mat_copy_csr_to_host(self._h, rp.ctypes.data_as(C.c_void_p),
                                                    ci.ctypes.data_as(C.c_void_p),
                                                    va.ctypes.data_as(C.c_void_p)))
        return rp, ci, va

    def CloneFrom(self, src):
        h = capi.mat_t()
        capi.check(_lib().ramd_mat_clone(src._h, C.byref(h)))
        _lib().ramd_mat_destroy(self._h)
        self._h = h
        self.dtype = src.dtype

    def CastFrom(self, src):
        h = capi.mat_t()
        capi.check(_lib().ramd_mat_cast(src._h, C.byref(h)))
        _lib().ramd_mat_destroy(self._h)
        self._h = h
        self.dtype = np.dtype(np.float32 if src.dtype == np.float64 else np.float64)

    # ---- unsmoothed-aggregation AMG setup, PMIS coarsening (local_matrix.cpp:6519-6640, :6852-6930)
    def AMGPMISAggregate(self, eps):
        """-> (connections, aggregates, aggregate_root_nodes) as int LocalVectors"""
        conn, agg, roots = LocalVector(np.int32), LocalVector(np.int32), LocalVector(np.int32)
        capi.check(_lib().ramd_mat_amg_pmis_aggregate(self._h, float(eps), conn._h, agg._h, roots._h))
        return conn, agg, roots

    def RSPMISCoarsening(self, eps):
        """-> (CFmap, S) int LocalVectors: 1 coarse / 2 fine per row, strong influence flag per entry"""
        cf, S = LocalVector(np.int32), LocalVector(np.int32)
        capi.check(_offscope("ramd_mat_rs_pmis_coarsening")(self._h, C.c_float(eps), cf._h, S._h))
        return cf, S

    def RSDirectInterpolation(self, CFmap, S, prolong):
        capi.check(_offscope("ramd_mat_rs_direct_interpolation")(self._h, CFmap._h, S._h, prolong._h))

    def AMGGreedyAggregate(self, eps):
        """-> (connections, aggregates, aggregate_root_nodes): the reference's sequential greedy sweep, same result"""
        conn, agg, roots = LocalVector(np.int32), LocalVector(np.int32), LocalVector(np.int32)
        capi.check(_lib().ramd_mat_amg_greedy_aggregate(self._h, float(eps), conn._h, agg._h, roots._h))
        return conn, agg, roots

    def AMGUnsmoothedAggregation(self, aggregates, aggregate_root_nodes, prolong):
        capi.check(_lib().ramd_mat_amg_unsmoothed_prolong(self._h, aggregates._h, aggregate_root_nodes._h, prolong._h))

    def AMGSmoothedAggregation(self, relax, connections, aggregates, aggregate_root_nodes, prolong, lumping_strat=0):
        capi.check(_lib().ramd_mat_amg_smoothed_prolong(self._h, float(relax), int(lumping_strat), connections._h,
                                                        aggregates._h, aggregate_root_nodes._h, prolong._h))

    def SPAI(self):
        """this becomes the sparse approximate inverse on its own pattern (host_matrix_csr.cpp:6665-6780)"""
        capi.check(_offscope("ramd_mat_spai")(self._h))

    def FSAI(self, power=1, pattern=None):
        """this becomes the FSAI factor on the lower pattern of A^power, or of `pattern` (host_matrix_csr.cpp:6514-6662)"""
        if pattern is not None:
            capi.check(_offscope("ramd_mat_fsai_pattern")(self._h, pattern._h))
        else:
            capi.check(_offscope("ramd_mat_fsai")(self._h, int(power)))

    def TripleMatrixProduct(self, R, A, P):
        tmp = LocalMatrix(self.dtype)
        tmp.MatrixMult(R, A)
        self.MatrixMult(tmp, P)

    # ---- CSR matrix algebra (host_matrix_csr.cpp Sort / Transpose / MatrixAdd / MatMatMult)
    def Sort(self):
        capi.check(_lib().ramd_mat_sort(self._h))

    def Transpose(self, out=None):
        """out = self^T (out given) or in place"""
        if out is not None:
            capi.check(_lib().ramd_mat_transpose(self._h, out._h))
            return
        tmp = LocalMatrix(self.dtype); tmp.CloneFrom(self)
        capi.check(_lib().ramd_mat_transpose(tmp._h, self._h))

    def MatrixAdd(self, mat, alpha=1.0, beta=1.0, structure=False):
        capi.check(_lib().ramd_mat_matrix_add(self._h, mat._h, float(alpha), float(beta), int(bool(structure))))

    def MatrixMult(self, A, B):
        capi.check(_lib().ramd_mat_mat_mult(self._h, A._h, B._h))

    def ConvertTo(self, fmt):
        """LocalMatrix::ConvertTo (src/base/local_matrix.cpp:2064-2151): a refused ELL conversion
        leaves the matrix in CSR (level-2 warning in the reference); returns the resulting format."""
        cur = self.GetFormat()
        if cur != CSR and int(fmt) != CSR and int(fmt) != cur:  # X -> CSR -> Y (local_matrix.cpp:2085-2093)
            capi.check(_lib().ramd_mat_convert(self._h, CSR))
        s = _lib().ramd_mat_convert(self._h, int(fmt))
        if s == capi.ERR_REFUSED:
            return self.GetFormat()
        capi.check(s)
        return self.GetFormat()

    def ConvertToCSR(self):
        return self.ConvertTo(CSR)

    def ConvertToELL(self):
        return self.ConvertTo(ELL)

    def ConvertToHYB(self):
        return self.ConvertTo(HYB)

    def ConvertToCOO(self):
        return self.ConvertTo(COO)

    def ell_arrays(self):
        w, c = C.c_int(0), C.c_int64(0)
        capi.check(_lib().ramd_mat_ell_info(self._h, C.byref(w), C.byref(c)))
        n = self.GetM()
        ec = np.empty(w.value * n, dtype=np.int32)
        ev = np.empty(w.value * n, dtype=self.dtype)
        capi.check(_lib().ramd_mat_copy_ell_to_host(self._h, ec.ctypes.data_as(C.c_void_p),
                                                    ev.ctypes.data_as(C.c_void_p)))
        return w.value, ec, ev

    def coo_arrays(self):
        w, c = C.c_int(0), C.c_int64(0)
        capi.check(_lib().ramd_mat_ell_info(self._h, C.byref(w), C.byref(c)))
        r = np.empty(c.value, dtype=np.int32)
        k = np.empty(c.value, dtype=np.int32)
        v = np.empty(c.value, dtype=self.dtype)
        capi.check(_lib().ramd_mat_copy_coo_to_host(self._h, r.ctypes.data_as(C.c_void_p),
                                                    k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        return r, k, v

    def Apply(self, x, y):
        capi.check(_lib().ramd_mat_apply(self._h, x._h, y._h))

    def UseRowPatterns(self, on=True):
        """False: products read the stored columns even where a row-pattern dictionary exists (ramd_mat_pattern_use)"""
        capi.check(_lib().ramd_mat_pattern_use(self._h, 1 if on else 0))

    def ApplyAdd(self, x, scalar, y):
        capi.check(_lib().ramd_mat_apply_add(self._h, x._h, float(scalar), y._h))

    def ExtractDiagonal(self, d):
        capi.check(_lib().ramd_mat_extract_diag(self._h, d._h))

    def ExtractInverseDiagonal(self, d):
        capi.check(_lib().ramd_mat_extract_inv_diag(self._h, d._h))

    def Gershgorin(self):
        """-> (lambda_min, lambda_max): Gershgorin bounds as the reference computes them (both start at 0)"""
        lo, hi = C.c_double(0), C.c_double(0)
        capi.check(_offscope("ramd_mat_gershgorin")(self._h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def ExtractL(self, out, diag):
        capi.check(_lib().ramd_mat_extract_tri(self._h, out._h, 0, int(bool(diag))))

    def ExtractU(self, out, diag):
        capi.check(_lib().ramd_mat_extract_tri(self._h, out._h, 1, int(bool(diag))))

    def Scale(self, alpha):
        capi.check(_lib().ramd_mat_scale_values(self._h, float(alpha), 0))

    def ScaleDiagonal(self, alpha):
        capi.check(_lib().ramd_mat_scale_values(self._h, float(alpha), 1))

    def ScaleOffDiagonal(self, alpha):
        capi.check(_lib().ramd_mat_scale_values(self._h, float(alpha), 2))

    def AddScalar(self, alpha):
        capi.check(_lib().ramd_mat_add_scalar_values(self._h, float(alpha), 0))

    def AddScalarDiagonal(self, alpha):
        capi.check(_lib().ramd_mat_add_scalar_values(self._h, float(alpha), 1))

    def AddScalarOffDiagonal(self, alpha):
        capi.check(_lib().ramd_mat_add_scalar_values(self._h, float(alpha), 2))

    def UpdateValuesCSR(self, val):
        va = np.ascontiguousarray(val, dtype=self.dtype)
        assert len(va) == self.GetNnz()
        capi.check(_lib().ramd_mat_update_values(self._h, va.ctypes.data_as(C.c_void_p)))

    def ExtractSubMatrix(self, row_offset, col_offset, row_size, col_size, out):
        capi.check(_lib().ramd_mat_extract_submatrix(self._h, row_offset, col_offset, row_size, col_size,
                                                     out._h))

    def Permute(self, perm):
        capi.check(_lib().ramd_mat_permute(self._h, perm._h))

    def MultiColoring(self):
        """-> (num_colors, size_colors, permutation LocalVector<int>)"""
        n = self.GetM()
        nc = C.c_int(0)
        sizes = np.zeros(max(n, 1), dtype=np.int32)
        perm = LocalVector(np.int32)
        capi.check(_lib().ramd_mat_multicoloring(self._h, C.byref(nc), sizes.ctypes.data_as(C.c_void_p),
                                                 perm._h))
        return nc.value, sizes[:nc.value].copy(), perm

    def ILU0Factorize(self):
        capi.check(_lib().ramd_mat_ilu0_factorize(self._h))

    def ILUpFactorize(self, p, level=True):
        capi.check(_lib().ramd_mat_ilup_factorize(self._h, int(p), 1 if level else 0))

    def LUAnalyse(self):
        capi.check(_lib().ramd_mat_lu_analyse(self._h))

    def LUAnalyseClear(self):
        capi.check(_lib().ramd_mat_lu_analyse_clear(self._h))

    def LUSolve(self, b, x):
        capi.check(_lib().ramd_mat_lu_solve(self._h, b._h, x._h))

    def LAnalyse(self, diag_unit=False):
        capi.check(_lib().ramd_mat_l_analyse(self._h, int(diag_unit)))

    def LAnalyseClear(self):
        capi.check(_lib().ramd_mat_l_analyse_clear(self._h))

    def LSolve(self, b, x):
        capi.check(_lib().ramd_mat_l_solve(self._h, b._h, x._h))

    def UAnalyse(self, diag_unit=False):
        capi.check(_lib().ramd_mat_u_analyse(self._h, int(diag_unit)))

    def UAnalyseClear(self):
        capi.check(_lib().ramd_mat_u_analyse_clear(self._h))

    def USolve(self, b, x):
        capi.check(_lib().ramd_mat_u_solve(self._h, b._h, x._h))

    def ReadFileMTX(self, filename):
        """LocalMatrix::ReadFileMTX with the reference's MatrixMarket semantics (include/rocalution/io.hpp)"""
        h = capi.mat_t()
        capi.check(_lib().ramd_mat_read_mtx(str(filename).encode(), _DT[self.dtype], C.byref(h)))
        _lib().ramd_mat_destroy(self._h)
        self._h = h

    def ReadFileCSR(self, filename):
        """LocalMatrix::ReadFileCSR: the reference's binary CSR file (host_io.cpp:497-609)"""
        h = capi.mat_t()
        capi.check(_lib().ramd_mat_read_file(str(filename).encode(), 1, _DT[self.dtype], C.byref(h)))
        _lib().ramd_mat_destroy(self._h)
        self._h = h

    def WriteFileMTX(self, filename):
        capi.check(_lib().ramd_mat_write_file(self._h, str(filename).encode(), 0))

    def WriteFileCSR(self, filename):
        capi.check(_lib().ramd_mat_write_file(self._h, str(filename).encode(), 1))

    def GenPoisson7(self, N):
        capi.check(_lib().ramd_mat_gen_poisson7(self._h, int(N)))

    def GenLaplace27(self, nx, ny=None, nz=None):
        """the reference's gen_3d_laplacian (clients/include/utility.hpp:110-177) on the device; ny, nz default to nx"""
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        capi.check(_lib().ramd_mat_gen_laplace27(self._h, int(nx), int(ny), int(nz)))
