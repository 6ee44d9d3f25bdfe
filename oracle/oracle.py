"""ORACLE -- TEST INFRASTRUCTURE ONLY (see the header of lsb_oracle.c).

ctypes/numpy front end of oracle/liblsb_oracle.so, plus a few numpy-only
restatements used to cross-check the C ones.  Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package
(lsbench_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build():
    """Compile the C restatement (and oracle/_ref when the reference is here)."""
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True)
    subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liblsb_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.orc_matrix_read.restype = C.c_void_p
    L.orc_matrix_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.orc_csr_free.argtypes = [C.c_void_p]
    L.orc_csr_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 3
    L.orc_csr_export.argtypes = [C.c_void_p, _u32p, _u32p, _f64p]
    L.orc_matrix_print_file.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_operator_upper.restype = C.c_uint32
    L.orc_operator_upper.argtypes = [C.c_uint32, C.c_uint32, _u32p, _u32p, _f64p,
                                     _u32p, _u32p, _f64p]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_get_max_threads.restype = C.c_int
    L.orc_spmv.argtypes = [C.c_uint64, _u64p, _u32p, _f64p, _f64p, _f64p]
    L.orc_pcg_jacobi.restype = C.c_int
    L.orc_pcg_jacobi.argtypes = [C.c_uint64, _u64p, _u32p, _f64p, _f64p, _f64p,
                                 C.c_double, C.c_uint32, C.c_int,
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    for name in ("orc_lap2d", "orc_lap3d", "orc_powerlaw", "orc_lap_coef"):
        getattr(L, name).restype = C.c_uint64
    L.orc_lap2d.argtypes = [C.c_uint64] * 4 + [C.c_void_p] * 3
    L.orc_lap3d.argtypes = [C.c_uint64] * 5 + [C.c_void_p] * 3
    L.orc_lap_coef.argtypes = [C.c_uint64] * 3 + [C.c_int] + [C.c_uint64] * 3 + [C.c_void_p] * 3
    L.orc_powerlaw_table.restype = C.c_double
    L.orc_powerlaw_table.argtypes = [C.c_double, C.c_uint32, _u64p]
    L.orc_powerlaw.argtypes = [C.c_uint64, C.c_uint32, _u64p, C.c_uint64,
                               C.c_uint64, C.c_uint64] + [C.c_void_p] * 3
    _LIB = L
    return L


class Csr:
    """0-based-offset CSR as plain numpy arrays (cols keep `base`)."""

    def __init__(self, nrows, base, offs, cols, vals):
        self.nrows, self.base = int(nrows), int(base)
        self.offs, self.cols, self.vals = offs, cols, vals

    @property
    def nnz(self):
        return int(self.offs[-1])

    def to_scipy(self):
        import scipy.sparse as sp
        ncols = max(self.nrows, int(self.cols.max()) - self.base + 1)
        return sp.csr_matrix((self.vals, self.cols.astype(np.int64) - self.base,
                              self.offs.astype(np.int64)), shape=(self.nrows, ncols))


class OracleError(Exception):
    pass


def matrix_read(path):
    """reference: src/lsbench-csr.c:29-92 (restated in lsb_oracle.c)."""
    L = lib()
    err = C.create_string_buffer(64)
    h = L.orc_matrix_read(os.fsencode(path), err, 64)
    if not h:
        raise OracleError(err.value.decode())
    n, b, z = C.c_uint32(), C.c_uint32(), C.c_uint32()
    L.orc_csr_dims(h, C.byref(n), C.byref(b), C.byref(z))
    offs = np.empty(n.value + 1, np.uint32)
    cols = np.empty(z.value, np.uint32)
    vals = np.empty(z.value, np.float64)
    L.orc_csr_export(h, offs, cols, vals)
    L.orc_csr_free(h)
    return Csr(n.value, b.value, offs, cols, vals)


def matrix_print(A, out_path):
    """reference: src/lsbench-csr.c:94-99, numpy side (format '%u %u %lf')."""
    with open(out_path, "w") as f:
        for i in range(A.nrows):
            for j in range(A.offs[i], A.offs[i + 1]):
                f.write("%u %u %f\n" % (i + A.base, A.cols[j], A.vals[j]))


def operator_upper(A):
    """S = triu(A) + triu(A,1)^T, the matrix CHOLMOD is handed
    (reference: src/cholmod-impl.h:5-21).  Returns a base-0 Csr."""
    L = lib()
    cap = 2 * A.nnz + 1
    so = np.zeros(A.nrows + 1, np.uint32)
    sc = np.zeros(cap, np.uint32)
    sv = np.zeros(cap, np.float64)
    z = L.orc_operator_upper(A.nrows, A.base, A.offs, A.cols, A.vals, so, sc, sv)
    return Csr(A.nrows, 0, so, sc[:z].copy(), sv[:z].copy())


def operator_upper_numpy(A):
    """Same operator through scipy.sparse, as a cross-check of the C one."""
    import scipy.sparse as sp
    M = A.to_scipy()[:, :A.nrows]
    U = sp.triu(M, 0, format="csr")
    S = (U + sp.triu(M, 1, format="csr").T).tocsr()
    S.sort_indices()
    return S


def rhs(n):
    """reference: src/lsbench.c:157-160 -- b_i = (double)i, 0-based row index."""
    return np.arange(n, dtype=np.float64)


def _as64(offs):
    return np.ascontiguousarray(offs, dtype=np.uint64)


def spmv(offs, cols, vals, x, threads=1):
    L = lib()
    L.orc_set_threads(threads)
    n = len(offs) - 1
    y = np.empty(n, np.float64)
    L.orc_spmv(n, _as64(offs), np.ascontiguousarray(cols, np.uint32),
               np.ascontiguousarray(vals, np.float64),
               np.ascontiguousarray(x, np.float64), y)
    return y


def pcg_jacobi(offs, cols, vals, b, tol=1e-12, maxit=20000, jacobi=True, threads=1):
    """Returns (x, iters, relres, status); status 1 converged, 2 breakdown, 3 maxit.
    jacobi: False/0 no preconditioner, True/1 M = diag(S), 2 M = diag(sum_j |S_ij|)."""
    L = lib()
    L.orc_set_threads(threads)
    n = len(offs) - 1
    x = np.zeros(n, np.float64)
    it, rel = C.c_uint32(), C.c_double()
    st = L.orc_pcg_jacobi(n, _as64(offs), np.ascontiguousarray(cols, np.uint32),
                          np.ascontiguousarray(vals, np.float64),
                          np.ascontiguousarray(b, np.float64), x, tol, maxit,
                          int(jacobi), C.byref(it), C.byref(rel))
    return x, it.value, rel.value, st


def pcg1_jacobi(offs, cols, vals, b, tol=1e-12, maxit=20000):
    """Single-reduction (Chronopoulos-Gear) Jacobi-PCG, sequential.
    Returns (x, iters, relres, status)."""
    L = lib()
    L.orc_set_threads(1)
    n = len(offs) - 1
    x = np.zeros(n, np.float64)
    it, rel = C.c_uint32(), C.c_double()
    L.orc_pcg1_jacobi.restype = C.c_int
    L.orc_pcg1_jacobi.argtypes = [C.c_uint64, _u64p, _u32p, _f64p, _f64p, _f64p, C.c_double,
                                  C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    st = L.orc_pcg1_jacobi(n, _as64(offs), np.ascontiguousarray(cols, np.uint32),
                           np.ascontiguousarray(vals, np.float64),
                           np.ascontiguousarray(b, np.float64), x, tol, maxit,
                           C.byref(it), C.byref(rel))
    return x, it.value, rel.value, st


def pcg_prec(offs, cols, vals, b, tol=1e-12, maxit=20000, kind="cheb", param=4):
    """PCG with z = M^-1 r as a vector operation: kind "cheb" = Chebyshev polynomial
    of degree `param` in D^-1 A, "bj" = block-Jacobi with `param`-row blocks (dense
    Cholesky per block), "fsai" = factorised sparse approximate inverse G^T G on the pattern of
    tril(A^param) (rows by dense Gaussian elimination).  Returns (x, iters, relres, status, spmvs, lmax)."""
    L = lib()
    L.orc_set_threads(1)
    n = len(offs) - 1
    x = np.zeros(n, np.float64)
    it, rel, nsp, lmax = C.c_uint32(), C.c_double(), C.c_uint32(), C.c_double()
    L.orc_pcg_prec.restype = C.c_int
    L.orc_pcg_prec.argtypes = [C.c_uint64, _u64p, _u32p, _f64p, _f64p, _f64p, C.c_double, C.c_uint32,
                               C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                               C.POINTER(C.c_uint32), C.POINTER(C.c_double)]
    st = L.orc_pcg_prec(n, _as64(offs), np.ascontiguousarray(cols, np.uint32),
                        np.ascontiguousarray(vals, np.float64), np.ascontiguousarray(b, np.float64),
                        x, tol, maxit, {"cheb": 3, "bj": 4, "fsai": 5}[kind], int(param), C.byref(it),
                        C.byref(rel), C.byref(nsp), C.byref(lmax))
    return x, it.value, rel.value, st, nsp.value, lmax.value


def gmres_jacobi(offs, cols, vals, b, tol=1e-10, maxit=20000, restart=30):
    """Restarted GMRES(m) with right Jacobi preconditioning (sequential).
    Returns (x, inner iterations, relres estimate, status)."""
    L = lib()
    L.orc_set_threads(1)
    n = len(offs) - 1
    x = np.zeros(n, np.float64)
    it, rel = C.c_uint32(), C.c_double()
    L.orc_gmres_jacobi.restype = C.c_int
    L.orc_gmres_jacobi.argtypes = [C.c_uint64, _u64p, _u32p, _f64p, _f64p, _f64p, C.c_double,
                                   C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_double)]
    st = L.orc_gmres_jacobi(n, _as64(offs), np.ascontiguousarray(cols, np.uint32),
                            np.ascontiguousarray(vals, np.float64),
                            np.ascontiguousarray(b, np.float64), x, tol, maxit, restart,
                            C.byref(it), C.byref(rel))
    return x, it.value, rel.value, st


def max_threads():
    return lib().orc_get_max_threads()


def _gen(fn, args, r0, r1):
    nnz = fn(*args, r0, r1, None, None, None)
    offs = np.empty(r1 - r0 + 1, np.uint64)
    cols = np.empty(nnz, np.uint32)
    vals = np.empty(nnz, np.float64)
    fn(*args, r0, r1, offs.ctypes.data, cols.ctypes.data, vals.ctypes.data)
    return offs, cols, vals


def lap2d(nx, ny, r0=0, r1=None):
    """5-point Laplacian, lexicographic, diag 4 / off-diag -1."""
    r1 = nx * ny if r1 is None else r1
    return _gen(lib().orc_lap2d, (nx, ny), r0, r1)


def lap3d(nx, ny, nz, r0=0, r1=None):
    """7-point Laplacian, lexicographic, diag 6 / off-diag -1."""
    r1 = nx * ny * nz if r1 is None else r1
    return _gen(lib().orc_lap3d, (nx, ny, nz), r0, r1)


def lap_coef(nx, ny, nz=None, coef=1, r0=0, r1=None):
    """Variable-coefficient 5-point (nz None) / 7-point operator, `...,coef=K`."""
    three = nz is not None
    n = nx * ny * (nz if three else 1)
    r1 = n if r1 is None else r1
    L = lib()
    return _gen(lambda *a: L.orc_lap_coef(nx, ny, nz if three else 1, int(three), coef, *a), (), r0, r1)


def powerlaw_gamma(avg, dmax):
    """Exponent for which the truncated discrete power law on [1,dmax] has the
    requested mean; bisection on the closed-form mean."""
    d = np.arange(1, dmax + 1, dtype=np.float64)

    def mean(g):
        w = d ** (-g)
        return float((d * w).sum() / w.sum())
    lo, hi = 0.0, 8.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if mean(mid) > avg:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def powerlaw_table(gamma, dmax):
    thr = np.empty(dmax, np.uint64)
    mean = lib().orc_powerlaw_table(gamma, dmax, thr)
    return thr, mean


def powerlaw(n, thr, seed, r0=0, r1=None):
    r1 = n if r1 is None else r1
    L = lib()
    dmax = len(thr)
    return _gen(lambda *a: L.orc_powerlaw(n, dmax, thr, seed, *a), (), r0, r1)
