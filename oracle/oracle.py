"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module (see oracle/README.md).
Every entry point wraps one function of krylov_oracle.c, which cites the
reference file:line it restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

CSR, DIA, ELL, HYB = 1, 5, 6, 7
CG, GMRES, BICGSTAB, FCG, CR, FGMRES, BICGSTABL, QMRCGSTAB, IDR, FIXEDPOINT, CHEBYSHEV = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
PC_NONE, PC_JACOBI, PC_ILU0, PC_MCSGS, PC_MCGS, PC_MCILU, PC_GS, PC_SGS, PC_IC = 0, 1, 2, 3, 4, 5, 6, 7, 8


def build(force=False):
    """Compile liboracle.so (and oracle/_ref/ref_probe when the image ships rocALUTION)."""
    src = [os.path.join(_HERE, f) for f in ("krylov_oracle.c", "krylov_oracle_impl.h", "krylov_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class SolveCfg(C.Structure):
    _fields_ = [
        ("solver", C.c_int), ("precond", C.c_int), ("format", C.c_int), ("basis", C.c_int),
        ("p0", C.c_double), ("p1", C.c_double),
        ("seed", C.c_ulonglong),
        ("abs_tol", C.c_double), ("rel_tol", C.c_double), ("div_tol", C.c_double),
        ("min_iter", C.c_int), ("max_iter", C.c_int),
        ("history", C.POINTER(C.c_double)), ("history_cap", C.c_int),
        ("iters", C.c_int), ("status", C.c_int),
        ("init_res", C.c_double), ("final_res", C.c_double), ("history_len", C.c_int),
        ("nblocks", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        for s in ("_f64", "_f32"):
            getattr(_lib, "orc_dot" + s).restype = C.c_double if s == "_f64" else C.c_float
            getattr(_lib, "orc_norm" + s).restype = C.c_double if s == "_f64" else C.c_float
            getattr(_lib, "orc_csr_extract_submatrix" + s).restype = C.c_int64
        _lib.orc_csr_hyb_coo_nnz.restype = C.c_int64
    return _lib


def set_threads(n):
    lib().orc_set_threads(int(n))


def set_solver_descr(iterative=False, max_iter=30, tol=1e-3, use_tol=True):
    """SolverDescr for the preconditioners built afterwards (TriSolverAlg_Iterative; solver.hpp:82-148)"""
    f = lib().orc_set_solver_descr
    f.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int]
    f.restype = None
    f(int(bool(iterative)), int(max_iter), float(tol), int(bool(use_tol)))


def max_threads():
    return lib().orc_max_threads()


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "_f64", C.c_double
    if dtype == np.float32:
        return "_f32", C.c_float
    raise TypeError(dtype)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fn(name, dtype):
    s, ct = _suf(dtype)
    return getattr(lib(), name + s), ct


# --------------------------------------------------------------------------- SpMV
def csr_apply(rp, ci, va, x):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_apply", va.dtype)
    y = np.zeros(len(rp) - 1, dtype=va.dtype)
    f(C.c_int(len(rp) - 1), _p(rp), _p(ci), _p(va), _p(np.ascontiguousarray(x)), _p(y))
    return y


def csr_apply_add(rp, ci, va, x, scalar, y):
    rp, ci = _i32(rp), _i32(ci)
    f, ct = _fn("orc_csr_apply_add", va.dtype)
    y = np.array(y, dtype=va.dtype, copy=True)
    f(C.c_int(len(rp) - 1), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), _p(np.ascontiguousarray(x)),
      ct(scalar), _p(y))
    return y


def csr_to_ell(rp, ci, va):
    """-> (width, ell_col, ell_val) or None when the reference refuses the conversion."""
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    w = lib().orc_csr_ell_width(C.c_int(n), C.c_int64(len(va)), _p(rp))
    if w < 0:
        return None
    ec = np.zeros(w * n, dtype=np.int32)
    ev = np.zeros(w * n, dtype=va.dtype)
    f, _ = _fn("orc_csr_to_ell_fill", va.dtype)
    f(C.c_int(n), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), C.c_int(w), _p(ec), _p(ev))
    return w, ec, ev


def csr_to_hyb(rp, ci, va):
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    w = lib().orc_csr_hyb_width(C.c_int(n), C.c_int64(len(va)))
    c = lib().orc_csr_hyb_coo_nnz(C.c_int(n), _p(rp), C.c_int(w))
    ec = np.zeros(w * n, dtype=np.int32)
    ev = np.zeros(w * n, dtype=va.dtype)
    cr = np.zeros(c, dtype=np.int32)
    cc = np.zeros(c, dtype=np.int32)
    cv = np.zeros(c, dtype=va.dtype)
    f, _ = _fn("orc_csr_to_hyb_fill", va.dtype)
    f(C.c_int(n), _p(rp), _p(ci), _p(va), C.c_int(w), _p(ec), _p(ev), _p(cr), _p(cc), _p(cv))
    return w, ec, ev, cr, cc, cv


def csr_transpose(rp, ci, va, ncol=None):
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    m = n if ncol is None else ncol
    f, _ = _fn("orc_csr_transpose", va.dtype)
    trp = np.zeros(m + 1, dtype=np.int32); tci = np.zeros(len(va), dtype=np.int32); tva = np.zeros(len(va), dtype=va.dtype)
    f(C.c_int(n), C.c_int(m), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), _p(trp), _p(tci), _p(tva))
    return trp, tci, tva


def csr_matmult(a, b, ncol_b=None):
    (arp, aci, ava), (brp, bci, bva) = a, b
    arp, aci, brp, bci = _i32(arp), _i32(aci), _i32(brp), _i32(bci)
    n = len(arp) - 1
    m = (len(brp) - 1) if ncol_b is None else ncol_b
    f, _ = _fn("orc_csr_matmult", ava.dtype)
    f.restype = C.c_int64
    crp = np.zeros(n + 1, dtype=np.int32)
    nnz = f(C.c_int(n), C.c_int(m), _p(arp), _p(aci), _p(ava), _p(brp), _p(bci), _p(bva), _p(crp), None, None)
    cci = np.zeros(nnz, dtype=np.int32); cva = np.zeros(nnz, dtype=ava.dtype)
    f(C.c_int(n), C.c_int(m), _p(arp), _p(aci), _p(ava), _p(brp), _p(bci), _p(bva), _p(crp), _p(cci), _p(cva))
    return crp, cci, cva


def csr_matrix_add(a, b, alpha, beta, structure):
    (arp, aci, ava), (brp, bci, bva) = a, b
    arp, aci, brp, bci = _i32(arp), _i32(aci), _i32(brp), _i32(bci)
    n = len(arp) - 1
    if not structure:
        f, ct = _fn("orc_csr_matrix_add_subset", ava.dtype)
        out = np.array(ava, copy=True)
        f(C.c_int(n), _p(arp), _p(aci), _p(out), _p(brp), _p(bci), _p(bva), ct(alpha), ct(beta))
        return arp, aci, out
    f, ct = _fn("orc_csr_matrix_add_union", ava.dtype)
    f.restype = C.c_int64
    crp = np.zeros(n + 1, dtype=np.int32)
    nnz = f(C.c_int(n), _p(arp), _p(aci), _p(ava), _p(brp), _p(bci), _p(bva), ct(alpha), ct(beta), _p(crp), None, None)
    cci = np.zeros(nnz, dtype=np.int32); cva = np.zeros(nnz, dtype=ava.dtype)
    f(C.c_int(n), _p(arp), _p(aci), _p(ava), _p(brp), _p(bci), _p(bva), ct(alpha), ct(beta), _p(crp), _p(cci), _p(cva))
    return crp, cci, cva


def csr_to_dia(rp, ci, va):
    """-> (offsets[num_diag], values[num_diag * n]) or None when the reference refuses the conversion."""
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    f, _ = _fn("orc_csr_to_dia", va.dtype)
    f.restype = C.c_int
    nd = f(C.c_int(n), C.c_int(n), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), None, None)
    if nd < 0:
        return None
    off = np.zeros(nd, dtype=np.int32)
    dv = np.zeros(nd * n, dtype=va.dtype)
    f(C.c_int(n), C.c_int(n), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), _p(off), _p(dv))
    return off, dv


def dia_apply(n, off, dv, x):
    f, _ = _fn("orc_dia_apply", dv.dtype)
    y = np.zeros(n, dtype=dv.dtype)
    f(C.c_int(n), C.c_int(len(off)), _p(_i32(off)), _p(dv), _p(np.ascontiguousarray(x, dtype=dv.dtype)), _p(y))
    return y


def dia_apply_add(n, off, dv, x, scalar, y):
    f, ct = _fn("orc_dia_apply_add", dv.dtype)
    y = np.array(y, dtype=dv.dtype, copy=True)
    f(C.c_int(n), C.c_int(len(off)), _p(_i32(off)), _p(dv), _p(np.ascontiguousarray(x, dtype=dv.dtype)), ct(scalar), _p(y))
    return y


def dia_to_csr(n, off, dv):
    f, _ = _fn("orc_dia_to_csr", dv.dtype)
    f.restype = C.c_int64
    rp = np.zeros(n + 1, dtype=np.int32)
    nnz = f(C.c_int(n), C.c_int(n), C.c_int(len(off)), _p(_i32(off)), _p(dv), _p(rp), None, None)
    ci = np.zeros(nnz, dtype=np.int32)
    va = np.zeros(nnz, dtype=dv.dtype)
    f(C.c_int(n), C.c_int(n), C.c_int(len(off)), _p(_i32(off)), _p(dv), _p(rp), _p(ci), _p(va))
    return rp, ci, va


def ell_apply(n, w, ec, ev, x):
    f, _ = _fn("orc_ell_apply", ev.dtype)
    y = np.zeros(n, dtype=ev.dtype)
    f(C.c_int(n), C.c_int(w), _p(_i32(ec)), _p(ev), _p(np.ascontiguousarray(x)), _p(y))
    return y


def ell_apply_add(n, w, ec, ev, x, scalar, y):
    f, ct = _fn("orc_ell_apply_add", ev.dtype)
    y = np.array(y, dtype=ev.dtype, copy=True)
    f(C.c_int(n), C.c_int(w), _p(_i32(ec)), _p(ev), _p(np.ascontiguousarray(x)), ct(scalar), _p(y))
    return y


def hyb_apply(n, ncol, w, ec, ev, cr, cc, cv, x):
    f, _ = _fn("orc_hyb_apply", ev.dtype)
    y = np.zeros(n, dtype=ev.dtype)
    f(C.c_int(n), C.c_int(ncol), C.c_int(w), _p(_i32(ec)), _p(ev), C.c_int64(len(cv)), _p(_i32(cr)),
      _p(_i32(cc)), _p(cv), _p(np.ascontiguousarray(x)), _p(y))
    return y


def hyb_apply_add(n, ncol, w, ec, ev, cr, cc, cv, x, scalar, y):
    f, ct = _fn("orc_hyb_apply_add", ev.dtype)
    y = np.array(y, dtype=ev.dtype, copy=True)
    f(C.c_int(n), C.c_int(ncol), C.c_int(w), _p(_i32(ec)), _p(ev), C.c_int64(len(cv)), _p(_i32(cr)),
      _p(_i32(cc)), _p(cv), _p(np.ascontiguousarray(x)), ct(scalar), _p(y))
    return y


def coo_apply(n, row, col, val, x):
    f, _ = _fn("orc_coo_apply", val.dtype)
    y = np.zeros(n, dtype=val.dtype)
    f(C.c_int(n), C.c_int64(len(val)), _p(_i32(row)), _p(_i32(col)), _p(val), _p(np.ascontiguousarray(x)),
      _p(y))
    return y


def coo_apply_add(row, col, val, x, scalar, y):
    f, ct = _fn("orc_coo_apply_add", val.dtype)
    y = np.array(y, dtype=val.dtype, copy=True)
    f(C.c_int64(len(val)), _p(_i32(row)), _p(_i32(col)), _p(val), _p(np.ascontiguousarray(x)), ct(scalar),
      _p(y))
    return y


# --------------------------------------------------------------------------- diagonal / BLAS-1
def extract_inv_diag(rp, ci, va):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_extract_inv_diag", va.dtype)
    d = np.zeros(len(rp) - 1, dtype=va.dtype)
    f(C.c_int(len(rp) - 1), _p(rp), _p(ci), _p(va), _p(d))
    return d


def extract_diag(rp, ci, va):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_extract_diag", va.dtype)
    d = np.zeros(len(rp) - 1, dtype=va.dtype)
    f(C.c_int(len(rp) - 1), _p(rp), _p(ci), _p(va), _p(d))
    return d


def add_scale(v, x, alpha):
    f, ct = _fn("orc_add_scale", v.dtype)
    v = v.copy()
    f(C.c_int64(len(v)), _p(v), _p(np.ascontiguousarray(x)), ct(alpha))
    return v


def scale_add(v, alpha, x):
    f, ct = _fn("orc_scale_add", v.dtype)
    v = v.copy()
    f(C.c_int64(len(v)), _p(v), ct(alpha), _p(np.ascontiguousarray(x)))
    return v


def scale_add2(v, alpha, x, beta, y, gamma):
    f, ct = _fn("orc_scale_add2", v.dtype)
    v = v.copy()
    f(C.c_int64(len(v)), _p(v), ct(alpha), _p(np.ascontiguousarray(x)), ct(beta),
      _p(np.ascontiguousarray(y)), ct(gamma))
    return v


def scale(v, alpha):
    f, ct = _fn("orc_scale", v.dtype)
    v = v.copy()
    f(C.c_int64(len(v)), _p(v), ct(alpha))
    return v


def dot(a, b):
    f, _ = _fn("orc_dot", a.dtype)
    return f(C.c_int64(len(a)), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)))


def norm(a):
    f, _ = _fn("orc_norm", a.dtype)
    return f(C.c_int64(len(a)), _p(np.ascontiguousarray(a)))


def pointwise_mult2(x, y):
    f, _ = _fn("orc_pointwise_mult2", x.dtype)
    v = np.zeros_like(x)
    f(C.c_int64(len(v)), _p(v), _p(np.ascontiguousarray(x)), _p(np.ascontiguousarray(y)))
    return v


def copy_permute(src, perm):
    f, _ = _fn("orc_copy_permute", src.dtype)
    dst = np.zeros_like(src)
    f(C.c_int64(len(src)), _p(dst), _p(np.ascontiguousarray(src)), _p(_i32(perm)))
    return dst


def copy_permute_backward(src, perm):
    f, _ = _fn("orc_copy_permute_backward", src.dtype)
    dst = np.zeros_like(src)
    f(C.c_int64(len(src)), _p(dst), _p(np.ascontiguousarray(src)), _p(_i32(perm)))
    return dst


# --------------------------------------------------------------------------- ILU / triangular / colouring
def ilu0(rp, ci, va):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_ilu0", va.dtype)
    lu = va.copy()
    f(C.c_int(len(rp) - 1), _p(rp), _p(ci), _p(lu))
    return lu


def symbolic_power(rp, ci, q):
    """pattern of A^q, rows sorted (host_matrix_csr.cpp:2718-2790 SymbolicMatMatMult, :3073-3146 SymbolicPower --
    beyond 8 the reference's loop multiplies once more)"""
    if q > 8:
        q += 1
    ones = np.ones(len(ci), dtype=np.float64)
    a = (_i32(rp), _i32(ci), ones)
    s = a
    for _ in range(q - 1):
        crp, cci, _v = csr_matmult(s, a)
        for i in range(len(crp) - 1):
            cci[crp[i]:crp[i + 1]].sort()
        s = (crp, cci, np.ones(len(cci), dtype=np.float64))
    return s[0], s[1]


def ilup(rp, ci, va, p, level=True):
    """LocalMatrix::ILUpFactorize (local_matrix.cpp:3910-4040) -> (rowptr, col, val) of the factors"""
    rp, ci = _i32(rp), _i32(ci)
    if p == 0:
        return rp, ci, ilu0(rp, ci, va)
    srp, sci = symbolic_power(rp, ci, p + 1)
    f, _ = _fn("orc_csr_ilup_numeric", va.dtype)
    val = np.zeros(len(sci), dtype=va.dtype)
    lev = np.zeros(len(sci), dtype=np.int32)
    # p = -1: nothing is eliminated, the call only places A's values on the pattern
    f(C.c_int(len(rp) - 1), C.c_int(p if level else -1), _p(srp), _p(sci), _p(rp), _p(ci), _p(va), _p(val), _p(lev))
    if not level:  # :3985-3993: values on the whole power pattern, then ILU(0)
        a0 = np.zeros(len(sci), dtype=va.dtype)
        row = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
        pos = {}
        srow = np.repeat(np.arange(len(srp) - 1), np.diff(srp))
        for k in range(len(sci)):
            pos[(int(srow[k]), int(sci[k]))] = k
        for k in range(len(ci)):
            a0[pos[(int(row[k]), int(ci[k]))]] = va[k]
        return srp, sci, ilu0(srp, sci, a0)
    keep = lev <= p
    nrp = np.zeros(len(srp), dtype=np.int32)
    srow = np.repeat(np.arange(len(srp) - 1), np.diff(srp))
    np.add.at(nrp, srow[keep] + 1, 1)
    return np.cumsum(nrp).astype(np.int32), sci[keep].copy(), val[keep].copy()


def lusolve(rp, ci, lu, b):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_lusolve", lu.dtype)
    x = np.zeros(len(rp) - 1, dtype=lu.dtype)
    f(C.c_int(len(rp) - 1), C.c_int64(len(lu)), _p(rp), _p(ci), _p(lu), _p(np.ascontiguousarray(b)), _p(x))
    return x


def lsolve(rp, ci, va, b, diag_unit):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_lsolve", va.dtype)
    x = np.zeros(len(rp) - 1, dtype=va.dtype)
    f(C.c_int(len(rp) - 1), _p(rp), _p(ci), _p(va), C.c_int(int(diag_unit)), _p(np.ascontiguousarray(b)),
      _p(x))
    return x


def usolve(rp, ci, va, b, diag_unit):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_usolve", va.dtype)
    x = np.zeros(len(rp) - 1, dtype=va.dtype)
    f(C.c_int(len(rp) - 1), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), C.c_int(int(diag_unit)),
      _p(np.ascontiguousarray(b)), _p(x))
    return x


def multicoloring(rp, ci):
    """-> (num_colors, size_colors[num_colors], perm[n])"""
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    nc = C.c_int(0)
    sizes = np.zeros(max(n, 1), dtype=np.int32)
    perm = np.zeros(n, dtype=np.int32)
    lib().orc_csr_multicoloring(C.c_int(n), C.c_int64(len(ci)), _p(rp), _p(ci), C.byref(nc), _p(sizes),
                                _p(perm))
    return nc.value, sizes[:nc.value].copy(), perm


def csr_permute(rp, ci, va, perm):
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    f, _ = _fn("orc_csr_permute", va.dtype)
    orp = np.zeros(n + 1, dtype=np.int32)
    oci = np.zeros(len(ci), dtype=np.int32)
    ova = np.zeros(len(va), dtype=va.dtype)
    f(C.c_int(n), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), _p(_i32(perm)), _p(orp), _p(oci), _p(ova))
    return orp, oci, ova


def extract_submatrix(rp, ci, va, r0, c0, rs, cs):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_csr_extract_submatrix", va.dtype)
    m = f(_p(rp), _p(ci), _p(va), C.c_int(r0), C.c_int(c0), C.c_int(rs), C.c_int(cs), None, None, None)
    orp = np.zeros(rs + 1, dtype=np.int32)
    oci = np.zeros(m, dtype=np.int32)
    ova = np.zeros(m, dtype=va.dtype)
    if m > 0:
        f(_p(rp), _p(ci), _p(va), C.c_int(r0), C.c_int(c0), C.c_int(rs), C.c_int(cs), _p(orp), _p(oci),
          _p(ova))
    return orp, oci, ova


def precond_apply(kind, rp, ci, va, rhs):
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_precond_apply", va.dtype)
    x = np.zeros(len(rp) - 1, dtype=va.dtype)
    f(C.c_int(kind), C.c_int(len(rp) - 1), C.c_int64(len(va)), _p(rp), _p(ci), _p(va),
      _p(np.ascontiguousarray(rhs, dtype=va.dtype)), _p(x))
    return x


def precond_apply_rep(kind, rp, ci, va, rhs, reps, x0=None):
    """`reps` applies of one built preconditioner, x carried over (starts from x0 or zeros)"""
    rp, ci = _i32(rp), _i32(ci)
    f, _ = _fn("orc_precond_apply_rep", va.dtype)
    x = np.zeros(len(rp) - 1, dtype=va.dtype) if x0 is None else np.array(x0, dtype=va.dtype)
    f(C.c_int(kind), C.c_int(len(rp) - 1), C.c_int64(len(va)), _p(rp), _p(ci), _p(va),
      _p(np.ascontiguousarray(rhs, dtype=va.dtype)), _p(x), C.c_int(reps))
    return x


# --------------------------------------------------------------------------- solvers
def _cfg(solver, precond, fmt, basis, abs_tol, rel_tol, div_tol, max_iter, min_iter, hist_cap):
    cfg = SolveCfg()
    cfg.solver, cfg.precond, cfg.format, cfg.basis = solver, precond, fmt, basis
    cfg.abs_tol, cfg.rel_tol, cfg.div_tol = abs_tol, rel_tol, div_tol
    cfg.min_iter, cfg.max_iter = min_iter, max_iter
    hist = np.zeros(hist_cap, dtype=np.float64) if hist_cap > 0 else None
    if hist is not None:
        cfg.history = hist.ctypes.data_as(C.POINTER(C.c_double))
        cfg.history_cap = hist_cap
    return cfg, hist


def solve(rp, ci, va, rhs, x0=None, solver=CG, precond=PC_NONE, fmt=CSR, basis=30, abs_tol=1e-15,
          rel_tol=1e-6, div_tol=1e8, max_iter=1000000, min_iter=0, history=True, hist_cap=None, seed=0, p0=0.0, p1=0.0,
          nblocks=1):
    """Build()+Solve() with the reference's control flow. Returns dict(x, iters, status, init_res,
    final_res, history).  nblocks = P > 1: the preconditioner is BlockJacobi over P contiguous row blocks (what a P-rank
    run of the reference computes, preconditioner_blockjacobi.cpp:80-141)."""
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    dtype = va.dtype
    x = np.zeros(n, dtype=dtype) if x0 is None else np.array(x0, dtype=dtype, copy=True)
    cap = (hist_cap if hist_cap is not None else min(max_iter + 2, 200000)) if history else 0
    cfg, hist = _cfg(solver, precond, fmt, basis, abs_tol, rel_tol, div_tol, max_iter, min_iter, cap)
    cfg.seed = int(seed)
    cfg.p0, cfg.p1 = float(p0), float(p1)
    cfg.nblocks = int(nblocks)
    f, _ = _fn("orc_solve", dtype)
    f(C.c_int(n), C.c_int64(len(va)), _p(rp), _p(ci), _p(va), _p(np.ascontiguousarray(rhs, dtype=dtype)),
      _p(x), C.byref(cfg))
    return dict(x=x, iters=cfg.iters, status=cfg.status, init_res=cfg.init_res, final_res=cfg.final_res,
                history=(hist[:min(cfg.history_len, cap)].copy() if hist is not None else None))


def solve_mixed(rp, ci, va, rhs, x0=None, outer=None, inner=None):
    """MixedPrecisionDC<double,float>: outer/inner are dicts of solve() keyword arguments."""
    rp, ci = _i32(rp), _i32(ci)
    n = len(rp) - 1
    x = np.zeros(n, dtype=np.float64) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    o = dict(solver=CG, precond=PC_NONE, fmt=CSR, basis=30, abs_tol=1e-15, rel_tol=1e-6, div_tol=1e8,
             max_iter=1000000, min_iter=0)
    o.update(outer or {})
    i = dict(solver=CG, precond=PC_NONE, fmt=CSR, basis=30, abs_tol=1e-15, rel_tol=1e-6, div_tol=1e8,
             max_iter=1000000, min_iter=0)
    i.update(inner or {})
    ocfg, hist = _cfg(o["solver"], o["precond"], o["fmt"], o["basis"], o["abs_tol"], o["rel_tol"],
                      o["div_tol"], o["max_iter"], o["min_iter"], 4096)
    icfg, _ = _cfg(i["solver"], i["precond"], i["fmt"], i["basis"], i["abs_tol"], i["rel_tol"],
                   i["div_tol"], i["max_iter"], i["min_iter"], 0)
    tot = C.c_int(0)
    lib().orc_solve_mixed(C.c_int(n), C.c_int64(len(va)), _p(rp), _p(ci),
                          _p(np.ascontiguousarray(va, dtype=np.float64)),
                          _p(np.ascontiguousarray(rhs, dtype=np.float64)), _p(x), C.byref(ocfg),
                          C.byref(icfg), C.byref(tot))
    return dict(x=x, iters=ocfg.iters, status=ocfg.status, init_res=ocfg.init_res,
                final_res=ocfg.final_res, history=hist[:ocfg.history_len].copy(),
                inner_iters=tot.value)
