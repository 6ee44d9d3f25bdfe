"""Thin Python handles over the C-ABI (include/lsbench.h, include/lsbench_hip.h).

Names follow the reference's C API (lsbench_matrix_read, lsbench_bench, ...;
reference: src/lsbench.h:32-40) so that tests read like a driver program.
Nothing is computed here: every method is one call into liblsbench_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(a):
    """Device or host address of a torch tensor / numpy array / int."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a)


class Matrix:
    """Owner of a `struct csr *` allocated by the library."""

    def __init__(self, handle):
        if not handle:
            raise L.LsbenchHipError("NULL struct csr")
        self._h = handle

    @classmethod
    def from_arrays(cls, offs, cols, vals, base=0):
        """Wrap numpy arrays as a struct csr (arrays are kept alive by self)."""
        self = cls.__new__(cls)
        self._offs = np.ascontiguousarray(offs, dtype=np.uint32)
        self._cols = np.ascontiguousarray(cols, dtype=np.uint32)
        self._vals = np.ascontiguousarray(vals, dtype=np.float64)
        s = L.CsrStruct(len(self._offs) - 1, base,
                        self._offs.ctypes.data_as(C.POINTER(C.c_uint)),
                        self._cols.ctypes.data_as(C.POINTER(C.c_uint)),
                        self._vals.ctypes.data_as(C.POINTER(C.c_double)))
        self._own = s
        self._h = C.pointer(s)
        return self

    @property
    def ptr(self):
        return self._h

    @property
    def nrows(self):
        return int(self._h.contents.nrows)

    @property
    def base(self):
        return int(self._h.contents.base)

    @property
    def nnz(self):
        return int(self._h.contents.offs[self.nrows])

    @property
    def offs(self):
        return np.ctypeslib.as_array(self._h.contents.offs, shape=(self.nrows + 1,))

    @property
    def cols(self):
        return np.ctypeslib.as_array(self._h.contents.cols, shape=(max(self.nnz, 1),))[:self.nnz]

    @property
    def vals(self):
        return np.ctypeslib.as_array(self._h.contents.vals, shape=(max(self.nnz, 1),))[:self.nnz]

    def free(self):
        if getattr(self, "_own", None) is None and self._h:
            L.load().lsb_csr_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def lsbench_matrix_read(path):
    return Matrix(L.load().lsbench_matrix_read(str(path).encode()))


def lsbench_matrix_synth(spec, r0=0, r1=0):
    n = C.c_uint()
    h = L.load().lsbench_matrix_synth(spec.encode(), r0, r1, C.byref(n))
    if not h:
        raise L.LsbenchHipError("bad synthetic spec %r" % spec)
    m = Matrix(h)
    m.n_global = n.value
    return m


def lsb_csr_symmetrize_upper(A):
    return Matrix(L.load().lsb_csr_symmetrize_upper(A.ptr))


def lsb_csr_copy_base0(A):
    return Matrix(L.load().lsb_csr_copy_base0(A.ptr))


def lsb_csr_row_slice(A, r0, r1):
    return Matrix(L.load().lsb_csr_row_slice(A.ptr, r0, r1))


def lsb_csr_partition_rows(A, nparts):
    b = (C.c_uint * (nparts + 1))()
    L.check(L.load().lsb_csr_partition_rows(A.ptr, nparts, b), "partition_rows")
    return np.array(b[:], dtype=np.int64)


def lsb_csr_row_blocks(A, cap):
    p = C.POINTER(C.c_uint)()
    nb = L.load().lsb_csr_row_blocks(A.ptr, cap, C.byref(p))
    out = np.ctypeslib.as_array(p, shape=(nb + 1,)).copy()
    C.CDLL(None).free(p)
    return out


def lsb_csr_block_lanes(A, rowblk):
    rb = np.ascontiguousarray(rowblk, dtype=np.uint32)
    out = np.zeros(len(rb) - 1, np.uint8)
    L.load().lsb_csr_block_lanes(A.ptr, rb.ctypes.data_as(C.POINTER(C.c_uint)), len(rb) - 1,
                                 out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def lsb_csr_sellize(A):
    """(sptr, cols, vals) of the sliced-ELL copy, as numpy arrays (copies)."""
    lib = L.load()
    p = lib.lsb_csr_sellize(A.ptr)
    if not p:
        raise L.LsbenchHipError("sliced-ELL copy does not fit 32-bit offsets")
    S = p.contents
    sptr = np.ctypeslib.as_array(S.sptr, (S.nslice + 1,)).copy()
    cols = np.ctypeslib.as_array(S.cols, (S.stored,)).copy()
    vals = np.ctypeslib.as_array(S.vals, (S.stored,)).copy()
    assert int(lib.lsb_csr_sell_stored(A.ptr)) == int(S.stored) == int(sptr[-1])
    lib.lsb_sell_free(p)
    return sptr, cols, vals


def lsb_csr_sellize16(A, row_begin=0):
    """(sptr, codes, sbase, vals) of the 16-bit sliced-ELL copy, or None when
    the operator cannot be encoded."""
    lib = L.load()
    p = lib.lsb_csr_sellize16(A.ptr, row_begin)
    if not p:
        return None
    S = p.contents
    sptr = np.ctypeslib.as_array(S.sptr, (S.nslice + 1,)).copy()
    nq = S.stored // L.SELL_ROWS
    codes = np.ctypeslib.as_array(S.codes, (max(S.ncode_slots, 1) * L.SELL_ROWS,)).copy()[
        :S.ncode_slots * L.SELL_ROWS]
    vals = np.ctypeslib.as_array(S.vals, (max(S.stored, 1),)).copy()[:S.stored]
    sbase = np.ctypeslib.as_array(S.sbase, (2 * nq + 2,)).copy()[:2 * nq].reshape(nq, 2)
    lib.lsb_sell_free(p)
    return sptr, codes, sbase, vals


def lsb_csr_rcm(A):
    perm = np.zeros(A.nrows, np.uint32)
    L.check(L.load().lsb_csr_rcm(A.ptr, perm.ctypes.data_as(C.POINTER(C.c_uint))), "rcm")
    return perm


def lsb_csr_permute_sym(A, perm):
    perm = np.ascontiguousarray(perm, dtype=np.uint32)
    return Matrix(L.load().lsb_csr_permute_sym(A.ptr, perm.ctypes.data_as(C.POINTER(C.c_uint))))


def lsb_csr_bandwidth(A):
    return int(L.load().lsb_csr_bandwidth(A.ptr))


def lsb_csr_col_hull(A):
    lo, hi = C.c_uint(), C.c_uint()
    L.load().lsb_csr_col_hull(A.ptr, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def lsb_plan_exchange(me, hull):
    """hull: (nall, 4) array of {row_begin, nrows, col_lo, col_hi}.
    Returns (recvs, sends) as lists of (peer, offset, count)."""
    hull = np.ascontiguousarray(hull, dtype=np.uint32)
    nall = hull.shape[0]
    recv, send = (L.Xfer * nall)(), (L.Xfer * nall)()
    nr, ns = C.c_int(), C.c_int()
    L.load().lsb_plan_exchange(me, nall, hull.ctypes.data_as(C.POINTER(C.c_uint)),
                               recv, C.byref(nr), send, C.byref(ns))
    f = lambda a, n: [(a[i].peer, a[i].offset, a[i].count) for i in range(n)]
    return f(recv, nr.value), f(send, ns.value)


def default_opts(**kw):
    o = L.Opts()
    L.load().lsb_hip_opts_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def hip_cdna4_init():
    return L.load().hip_cdna4_init()


def hip_cdna4_finalize():
    return L.load().hip_cdna4_finalize()


def hip_cdna4_bench(A, r=None, trials=1, matrix_name="(memory)", opts=None):
    """The drop-in entry point, called the way lsbench_bench calls a backend
    (reference: src/lsbench.c:156-187): x = zeros, r_i = i unless given.
    Returns x (numpy)."""
    lib = L.load()
    m = A.nrows
    x = np.zeros(m, np.float64)
    r = np.arange(m, dtype=np.float64) if r is None else np.ascontiguousarray(r, np.float64)
    if opts is not None:
        lib.lsb_hip_set_opts(C.byref(opts))
    cb = L.LsbenchStruct(matrix_name.encode(), L.SOLVER_HIP, 0, 0, 0, trials)
    rc = lib.hip_cdna4_bench(x.ctypes.data_as(C.POINTER(C.c_double)), A.ptr,
                             r.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cb))
    L.check(rc, "hip_cdna4_bench")
    return x


def last_result():
    r = L.Result()
    L.load().lsb_hip_last_result(C.byref(r))
    return r


class Solver:
    """lsb_hip_solver handle: operator resident in HBM."""

    def __init__(self, A, opts=None, row_begin=None, n_global=None):
        lib = L.load()
        self.opts = opts if opts is not None else default_opts()
        if row_begin is None:
            h = lib.lsb_hip_solver_create(A.ptr, C.byref(self.opts))
        else:
            h = lib.lsb_hip_solver_create_dist(A.ptr, row_begin, n_global, C.byref(self.opts))
        if not h:
            raise L.LsbenchHipError("lsb_hip_solver_create failed: backend not "
                                    "initialised (no GPU?) -- there is no CPU path")
        self._h = h
        self.n_local = lib.lsb_hip_solver_nrows_local(h)
        self.n_global = lib.lsb_hip_solver_nrows_global(h)
        self.nnz_local = lib.lsb_hip_solver_nnz_local(h)

    def solve(self, b):
        x = np.empty(self.n_local, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        res = L.Result()
        L.check(L.load().lsb_hip_solver_solve(self._h, b.ctypes.data, x.ctypes.data,
                                              C.byref(res)), "solve")
        return x, res

    def solve_dev(self, d_b, d_x):
        res = L.Result()
        L.check(L.load().lsb_hip_solver_solve_dev(self._h, _ptr(d_b), _ptr(d_x),
                                                  C.byref(res)), "solve_dev")
        return res

    def spmv_dev(self, d_x, d_y):
        L.check(L.load().lsb_hip_solver_spmv_dev(self._h, _ptr(d_x), _ptr(d_y)), "spmv_dev")

    def time_spmv(self, warm=5, reps=50):
        ms = C.c_double()
        L.check(L.load().lsb_hip_solver_time_spmv(self._h, warm, reps, C.byref(ms)),
                "time_spmv")
        return ms.value

    def jacobi_sweep_dev(self, w, d_b, d_x):
        L.check(L.load().lsb_hip_solver_jacobi_sweep_dev(self._h, w, _ptr(d_b), _ptr(d_x)),
                "jacobi_sweep_dev")

    @property
    def nblocks(self):
        return L.load().lsb_hip_solver_nblocks(self._h)

    @property
    def spmv_variant(self):
        return L.load().lsb_hip_solver_spmv_variant(self._h)

    @property
    def spmv_flags(self):
        return L.load().lsb_hip_solver_spmv_flags(self._h)

    @property
    def spmv_grid(self):
        return L.load().lsb_hip_solver_spmv_grid(self._h)

    @property
    def padded(self):
        """Pad rows of a line-padded 2-D grid inside the solver (0: none)."""
        return int(L.load().lsb_hip_solver_padded(self._h))

    @property
    def spmv_col_slices(self):
        """Slices shard 0's z-column plan walks in columns (k_spmv_tmpl_col); 0 = no plan."""
        return int(L.load().lsb_hip_solver_spmv_col_slices(self._h))

    @property
    def spmv_period(self):
        return L.load().lsb_hip_solver_spmv_period(self._h)

    @property
    def sell_value_slots(self):
        """(slots that keep their 128 values, all slots) of the 16-bit sliced-ELL form in use,
        (0, 0) for every other form."""
        import ctypes as C
        k, t = C.c_uint(0), C.c_uint(0)
        L.load().lsb_hip_solver_sell_value_slots(self._h, C.byref(k), C.byref(t))
        return int(k.value), int(t.value)

    @property
    def spmv_layout_bytes(self):
        """Bytes one launch of the SpMV form in use must move: its matrix-side arrays as
        stored + x once + y once (0 for the multi-pass forms)."""
        return int(L.load().lsb_hip_solver_spmv_layout_bytes(self._h))

    @property
    def iteration_bytes(self):
        """Bytes one Krylov iteration must move (SpMV layout + the sweeps' vector passes); 0 where
        the iteration has another shape (see lsbench_hip.h)."""
        return int(L.load().lsb_hip_solver_iteration_bytes(self._h))

    @property
    def single_reduction(self):
        """True where the PCG iteration runs in its single-reduction form (see lsbench_hip.h)."""
        return bool(L.load().lsb_hip_solver_single_reduction(self._h))

    @property
    def fused_p(self):
        """0 / 1 / 2: the direction update rides in the next SpMV launch (see lsbench_hip.h)."""
        return int(L.load().lsb_hip_solver_fused_p(self._h))

    @property
    def blas1_nt(self):
        """mask of the sweeps' nontemporal operands (see lsbench_hip.h)"""
        return int(L.load().lsb_hip_solver_blas1_nt(self._h))

    @property
    def comm_plan(self):
        """The exchange plan of this process's first shard (see lsbench_hip.h)."""
        p = (C.c_ulonglong * 8)()
        L.load().lsb_hip_solver_comm_plan(self._h, p)
        us = (C.c_double * 2)()
        L.load().lsb_hip_solver_overlap(self._h, us)
        return {"rccl_ranks": int(p[0]), "recv_peers": int(p[1]), "send_peers": int(p[2]),
                "bytes_recv_per_exchange": int(p[3]), "bytes_sent_per_exchange": int(p[4]),
                "pattern": "all-gather" if p[5] else "halos (point-to-point)",
                "shards_in_process": int(p[6]), "overlap": bool(p[7]),
                # opts.overlap = -1: both forms of the sharded SpMV timed on this communicator at creation
                "overlap_timed_us": {"plain": us[0], "split": us[1]} if us[0] > 0 else None}

    @property
    def comm(self):
        """(mode, p2p_us, rccl_us): mode 0 = one shard, 1 = RCCL / device copies,
        2 = direct xGMI stores for the all-reduces, 3 = and for the halos; the
        two times are the creation-time self-test's cost of one exchange +
        all-reduce each way (0 when it did not run)."""
        a, b = C.c_double(), C.c_double()
        m = L.load().lsb_hip_solver_comm(self._h, C.byref(a), C.byref(b))
        return m, a.value, b.value

    @property
    def overlaps(self):
        """True when the halo exchange runs behind the interior rows."""
        return bool(L.load().lsb_hip_solver_overlaps(self._h))

    def destroy(self):
        if self._h:
            L.load().lsb_hip_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
