"""ctypes binding of liblsbench_hip.so -- the C-ABI declared in include/*.h.

This is plumbing: every function below is one C symbol with its C signature.
There is no Python implementation of anything and no CPU fallback; when the
library is missing the import fails, and when there is no GPU the backend's
own return code 1 ("not initialised") surfaces as LsbenchHipError.
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "liblsbench_hip.so")    # the backend: the product
CORE_PATH = os.path.join(_CSRC, "liblsbench_core.so")  # stand-in for the reference's liblsbench (CLI, loader, dispatch)
CORE_SYMBOLS = ("lsbench_matrix_read", "lsbench_matrix_print", "lsbench_matrix_free", "lsbench_init",
                "lsbench_get_matrix_name", "lsbench_bench", "lsbench_finalize")

UNIQUE_ID_BYTES = 128

# enums of include/lsbench.h / include/lsbench_hip.h
SOLVER_HIP = 6
OP_CHOLMOD_UPPER, OP_RAW = 0, 1
PRECOND_JACOBI, PRECOND_NONE, PRECOND_L1JACOBI, PRECOND_CHEBYSHEV, PRECOND_BLOCKJACOBI, PRECOND_FSAI = 0, 1, 2, 3, 4, 5
KRYLOV_PCG, KRYLOV_GMRES, KRYLOV_PCG1, KRYLOV_AUTO = 0, 1, 2, 3
SPMV_AUTO, SPMV_ADAPTIVE, SPMV_SUBWAVE, SPMV_SCALAR, SPMV_PANEL, SPMV_SELL, SPMV_BINNED, SPMV_TWOPHASE = 0, 1, 2, 3, 4, 5, 6, 7
SELL_ROWS = 128
BIN_CHUNK = 2048
PB_COLS, PB_ROWS = 8192, 2048
SPMV_FLAG_PREFETCH, SPMV_FLAG_NT, SPMV_FLAG_C16, SPMV_FLAG_TMPL, SPMV_FLAG_DEFER, SPMV_FLAG_COL = 1, 2, 4, 64, 128, 256
STATUS_RUNNING, STATUS_CONVERGED, STATUS_BREAKDOWN, STATUS_MAXIT = 0, 1, 2, 3
STATUS_COMM = 4
COMM_AUTO, COMM_RCCL, COMM_P2P = 0, 1, 2
PREC_FP64, PREC_MIXED = 0, 1


class LsbenchHipError(RuntimeError):
    pass


class CsrStruct(C.Structure):
    """struct csr (reference layout, src/lsbench-impl.h:22-26)."""
    _fields_ = [("nrows", C.c_uint), ("base", C.c_uint),
                ("offs", C.POINTER(C.c_uint)), ("cols", C.POINTER(C.c_uint)),
                ("vals", C.POINTER(C.c_double))]


class LsbenchStruct(C.Structure):
    """struct lsbench (reference layout, src/lsbench-impl.h:14-20)."""
    _fields_ = [("matrix", C.c_char_p), ("solver", C.c_int), ("ordering", C.c_int),
                ("precision", C.c_int), ("verbose", C.c_uint), ("trials", C.c_uint)]


class Opts(C.Structure):
    """struct lsb_hip_opts."""
    _fields_ = [("tol", C.c_double), ("maxit", C.c_uint), ("op_mode", C.c_int),
                ("precond", C.c_int), ("spmv_variant", C.c_int),
                ("check_every", C.c_int), ("use_graph", C.c_int),
                ("sample_spmv", C.c_int), ("nvirt", C.c_int), ("comm", C.c_int), ("overlap", C.c_int),
                ("spmv_tune", C.c_int), ("spmv_grid", C.c_int), ("reorder", C.c_int),
                ("krylov", C.c_int),
                ("restart", C.c_int), ("verbose", C.c_int),
                ("ngpus", C.c_int), ("verify", C.c_int), ("cheb_degree", C.c_int),
                ("block_size", C.c_int), ("precision", C.c_int), ("persistent", C.c_int),
                ("comm_deadline_s", C.c_double), ("fsai_power", C.c_int), ("blas1_nt", C.c_int)]


class Result(C.Structure):
    """struct lsb_hip_result."""
    _fields_ = [("iters", C.c_uint), ("status", C.c_int), ("relres", C.c_double),
                ("seconds", C.c_double), ("spmv_ms", C.c_double),
                ("spmv_samples", C.c_uint), ("corrections", C.c_uint),
                ("true_relres", C.c_double), ("spmvs", C.c_uint)]


class PanelCsr(C.Structure):
    """struct lsb_panel_csr."""
    _fields_ = [("npanels", C.c_uint), ("width", C.c_uint), ("npairs", C.c_uint),
                ("nrows", C.c_uint), ("pair_begin", C.POINTER(C.c_uint)),
                ("pair_row", C.POINTER(C.c_uint)), ("offs", C.POINTER(C.c_uint)),
                ("cols", C.POINTER(C.c_uint)), ("vals", C.POINTER(C.c_double))]


class Sell(C.Structure):
    """struct lsb_sell."""
    _fields_ = [("nrows", C.c_uint), ("nslice", C.c_uint), ("stored", C.c_ulonglong),
                ("sptr", C.POINTER(C.c_uint)), ("cols", C.POINTER(C.c_int)),
                ("vals", C.POINTER(C.c_double)), ("codes", C.POINTER(C.c_short)),
                ("sbase", C.POINTER(C.c_int)), ("ncode_slots", C.c_uint)]


class SellVc(C.Structure):
    """struct lsb_sell_vc."""
    _fields_ = [("nslots", C.c_ulonglong), ("nval_slots", C.c_uint), ("slots", C.POINTER(C.c_int)),
                ("vconst", C.POINTER(C.c_double)), ("vals", C.POINTER(C.c_double))]


class FsaiPattern(C.Structure):
    """struct lsb_fsai_pattern."""
    _fields_ = [("n", C.c_uint), ("cap", C.c_uint), ("nnz", C.c_ulonglong), ("offs", C.POINTER(C.c_uint)),
                ("cols", C.POINTER(C.c_uint))]


class SellTmpl(C.Structure):
    """struct lsb_sell_tmpl (176 bytes)."""
    _fields_ = [("nslots", C.c_int), ("shaped", C.c_int), ("base", C.c_int * 8), ("kidx", C.c_int * 8),
                ("kind", C.c_int * 8), ("pad_", C.c_int * 2), ("cst", C.c_double * 8)]


class SellTmpls(C.Structure):
    """struct lsb_sell_tmpls."""
    _fields_ = [("nslice", C.c_uint), ("ntmpl", C.c_uint), ("nfar", C.c_uint), ("covered", C.c_ulonglong),
                ("shaped", C.c_ulonglong), ("nmask", C.c_ulonglong), ("kept_read", C.c_ulonglong),
                ("tid", C.POINTER(C.c_ubyte)), ("vbase", C.POINTER(C.c_uint)),
                ("mask", C.POINTER(C.c_ulonglong)), ("t", C.POINTER(SellTmpl))]


class TmplCols(C.Structure):
    """struct lsb_tmpl_cols."""
    _fields_ = [("nitem", C.c_uint), ("kmax", C.c_uint), ("period", C.c_uint), ("s_lo", C.c_uint), ("s_hi", C.c_uint),
                ("xbeg", C.c_uint * 9),
                ("item", C.POINTER(C.c_uint)), ("in_cols", C.c_ulonglong), ("centre0", C.c_int)]


class Binned(C.Structure):
    """struct lsb_binned."""
    _fields_ = [("nbins", C.c_uint), ("width", C.c_uint), ("nrows", C.c_uint), ("nchunks", C.c_uint),
                ("chunk_cap", C.c_uint), ("nnz", C.c_ulonglong), ("bin_chunk", C.POINTER(C.c_uint)),
                ("chunk_begin", C.POINTER(C.c_uint)), ("rows", C.POINTER(C.c_uint)),
                ("cols", C.POINTER(C.c_uint)), ("vals", C.POINTER(C.c_double))]


class Pb(C.Structure):
    """struct lsb_pb."""
    _fields_ = [("nrows", C.c_uint), ("ncols_lo", C.c_uint), ("nchunks", C.c_uint), ("nbins", C.c_uint),
                ("nitems", C.c_uint), ("npieces", C.c_uint), ("cols", C.c_uint), ("rows", C.c_uint),
                ("nnz", C.c_ulonglong), ("nent", C.c_ulonglong),
                ("vals", C.POINTER(C.c_double)), ("colw", C.POINTER(C.c_ushort)),
                ("grp_first", C.POINTER(C.c_uint)), ("grp_mask", C.POINTER(C.c_ulonglong)),
                ("delta", C.POINTER(C.c_uint)), ("item", C.POINTER(C.c_uint)),
                ("bin_ptr", C.POINTER(C.c_uint)), ("roww", C.POINTER(C.c_ushort))]


class Xfer(C.Structure):
    """struct lsb_xfer."""
    _fields_ = [("peer", C.c_int), ("offset", C.c_size_t), ("count", C.c_size_t)]


_vp, _u, _i, _d = C.c_void_p, C.c_uint, C.c_int, C.c_double
_csrp = C.POINTER(CsrStruct)

# name -> (restype, argtypes): every symbol include/*.h declares
SIGNATURES = {
    # include/lsbench.h
    "lsbench_matrix_read": (_csrp, [C.c_char_p]),
    "lsbench_matrix_print": (None, [_csrp]),
    "lsbench_matrix_free": (None, [_csrp]),
    "lsbench_init": (C.POINTER(LsbenchStruct), [_i, C.POINTER(C.c_char_p)]),
    "lsbench_get_matrix_name": (C.c_char_p, [C.POINTER(LsbenchStruct)]),
    "lsbench_bench": (None, [_csrp, C.POINTER(LsbenchStruct)]),
    "lsbench_finalize": (None, [C.POINTER(LsbenchStruct)]),
    # include/lsbench_hip.h -- backend trio
    "hip_cdna4_init": (_i, []),
    "hip_cdna4_finalize": (_i, []),
    "hip_cdna4_bench": (_i, [C.POINTER(C.c_double), _csrp, C.POINTER(C.c_double),
                             C.POINTER(LsbenchStruct)]),
    # options / results
    "lsb_hip_opts_default": (None, [C.POINTER(Opts)]),
    "lsb_hip_set_opts": (None, [C.POINTER(Opts)]),
    "lsb_hip_get_opts": (None, [C.POINTER(Opts)]),
    "lsb_hip_last_result": (None, [C.POINTER(Result)]),
    "lsb_hip_device_count": (_i, []),
    # host helpers
    "lsb_csr_free": (None, [_csrp]),
    "lsb_csr_symmetrize_upper": (_csrp, [_csrp]),
    "lsb_csr_copy_base0": (_csrp, [_csrp]),
    "lsb_csr_row_slice": (_csrp, [_csrp, _u, _u]),
    "lsb_csr_partition_rows": (_i, [_csrp, _u, C.POINTER(_u)]),
    "lsb_csr_row_blocks": (_u, [_csrp, _u, C.POINTER(C.POINTER(_u))]),
    "lsb_csr_block_lanes": (None, [_csrp, C.POINTER(_u), _u, C.POINTER(C.c_ubyte)]),
    "lsb_csr_rcm": (_i, [_csrp, C.POINTER(_u)]),
    "lsb_csr_permute_sym": (_csrp, [_csrp, C.POINTER(_u)]),
    "lsb_csr_bandwidth": (_u, [_csrp]),
    "lsb_csr_panelize": (C.POINTER(PanelCsr), [_csrp, _u]),
    "lsb_panel_csr_free": (None, [C.POINTER(PanelCsr)]),
    "lsb_csr_pbize": (C.POINTER(Pb), [_csrp]),
    "lsb_csr_pbize2": (C.POINTER(Pb), [_csrp, _u, _u]),
    "lsb_pb_free": (None, [C.POINTER(Pb)]),
    "lsb_pb_check": (_i, [C.POINTER(Pb), C.c_ulonglong, C.c_ulonglong, _i, C.c_char_p, C.c_size_t]),
    "lsb_csr_binize": (C.POINTER(Binned), [_csrp, _u]),
    "lsb_binned_free": (None, [C.POINTER(Binned)]),
    "lsb_csr_sell_stored": (C.c_ulonglong, [_csrp]),
    "lsb_csr_sellize": (C.POINTER(Sell), [_csrp]),
    "lsb_csr_sellize16": (C.POINTER(Sell), [_csrp, _u]),
    "lsb_sell_free": (None, [C.POINTER(Sell)]),
    "lsb_sell16_value_slots": (C.POINTER(SellVc), [C.POINTER(Sell)]),
    "lsb_sell_vc_free": (None, [C.POINTER(SellVc)]),
    "lsb_csr_fsai_pattern": (C.POINTER(FsaiPattern), [_csrp, _i, _u]),
    "lsb_fsai_pattern_free": (None, [C.POINTER(FsaiPattern)]),
    "lsb_sell16_templates": (C.POINTER(SellTmpls), [C.POINTER(Sell), C.POINTER(SellVc)]),
    "lsb_sell_tmpls_free": (None, [C.POINTER(SellTmpls)]),
    "lsb_tmpl_check": (_i, [C.POINTER(Sell), C.POINTER(SellVc), C.POINTER(SellTmpls), _u, _u, _u, _i,
                            C.c_char_p, C.c_size_t]),
    "lsb_sell_tmpl_columns": (C.POINTER(TmplCols), [C.POINTER(SellTmpls), _u, _u]),
    "lsb_sell_tmpl_columns_range": (C.POINTER(TmplCols), [C.POINTER(SellTmpls), _u, _u, _u, _u]),
    "lsb_tmpl_cols_free": (None, [C.POINTER(TmplCols)]),
    "lsb_tmpl_cols_check": (_i, [C.POINTER(SellTmpls), C.POINTER(TmplCols), C.c_char_p, C.c_size_t]),
    "lsb_csr_mean_scatter": (_d, [_csrp, _u]),
    "lsb_csr_col_hull": (None, [_csrp, C.POINTER(_u), C.POINTER(_u)]),
    "lsb_plan_exchange": (None, [_i, _i, C.POINTER(_u), C.POINTER(Xfer), C.POINTER(_i),
                                 C.POINTER(Xfer), C.POINTER(_i)]),
    "lsbench_matrix_synth": (_csrp, [C.c_char_p, _u, _u, C.POINTER(_u)]),
    # solver handle
    "lsb_hip_solver_create": (_vp, [_csrp, C.POINTER(Opts)]),
    "lsb_hip_solver_create_dist": (_vp, [_csrp, _u, _u, C.POINTER(Opts)]),
    "lsb_hip_solver_destroy": (None, [_vp]),
    "lsb_hip_solver_solve": (_i, [_vp, _vp, _vp, C.POINTER(Result)]),
    "lsb_hip_solver_solve_dev": (_i, [_vp, _vp, _vp, C.POINTER(Result)]),
    "lsb_hip_solver_spmv_dev": (_i, [_vp, _vp, _vp]),
    "lsb_hip_solver_time_spmv": (_i, [_vp, _i, _i, C.POINTER(_d)]),
    "lsb_hip_solver_jacobi_sweep_dev": (_i, [_vp, _d, _vp, _vp]),
    "lsb_hip_solver_nrows_local": (_u, [_vp]),
    "lsb_hip_solver_padded": (_i, [_vp]),
    "lsb_csr_pad_lines": (_csrp, [_csrp, _u, C.POINTER(_u), C.POINTER(_u), C.POINTER(C.POINTER(C.c_int))]),
    "lsb_hip_solver_nrows_global": (_u, [_vp]),
    "lsb_hip_solver_nnz_local": (C.c_ulonglong, [_vp]),
    "lsb_hip_solver_nblocks": (_u, [_vp]),
    "lsb_hip_solver_spmv_variant": (_i, [_vp]),
    "lsb_hip_solver_spmv_flags": (_u, [_vp]),
    "lsb_hip_solver_spmv_grid": (_u, [_vp]),
    "lsb_hip_solver_spmv_period": (_u, [_vp]),
    "lsb_hip_solver_spmv_col_slices": (C.c_ulonglong, [_vp]),
    "lsb_hip_solver_sell_value_slots": (None, [_vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "lsb_hip_solver_overlaps": (_i, [_vp]),
    "lsb_hip_solver_overlap": (_i, [_vp, C.POINTER(C.c_double)]),
    "lsb_hip_solver_comm_plan": (None, [_vp, C.POINTER(C.c_ulonglong)]),
    "lsb_hip_solver_spmv_layout_bytes": (C.c_ulonglong, [_vp]),
    "lsb_hip_solver_fused_p": (_i, [_vp]),
    "lsb_hip_solver_iteration_bytes": (C.c_ulonglong, [_vp]),
    "lsb_hip_solver_single_reduction": (_i, [_vp]),
    "lsb_hip_solver_blas1_nt": (_i, [_vp]),
    "lsb_hip_solver_comm": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "lsb_hip_stream": (_vp, []),
    # communicator
    "lsb_hip_comm_get_unique_id": (_i, [_vp]),
    "lsb_hip_comm_init_rank": (_i, [_vp, _i, _i]),
    "lsb_hip_comm_destroy": (_i, []),
    "lsb_hip_comm_rank": (_i, []),
    "lsb_hip_comm_size": (_i, []),
    "lsb_hip_comm_count": (_i, []),
    "hip_cdna4_set_option": (_i, [C.c_char_p, C.c_char_p]),
    "hip_cdna4_matrix_synth": (_csrp, [C.c_char_p]),
    "lsb_hip_comm_allreduce_sum_dev": (_i, [_vp, _i]),
    "lsb_hip_comm_barrier": (_i, []),
    # kernel-level entry points
    "lsb_hip_spmv_csr_f64": (_i, [_i, _u, _vp, _vp, _vp, _vp, _vp, _u, _u, _u, _vp, _vp,
                                  _vp, _vp, _vp, _vp]),
    "lsb_hip_partials_capacity": (_u, []),
    "lsb_hip_dot_f64": (_i, [_u, _vp, _vp, _vp, _vp, _vp]),
    "lsb_hip_nrm2_f64": (_i, [_u, _vp, _vp, _vp, _vp]),
    "lsb_hip_axpy_f64": (_i, [_u, _vp, _vp, _vp, _vp]),
    "lsb_hip_xpay_f64": (_i, [_u, _vp, _vp, _vp, _vp]),
    "lsb_hip_jacobi_setup_f64": (_i, [_u, _u, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lsb_hip_jacobi_apply_f64": (_i, [_u, _vp, _vp, _vp, _vp]),
    "lsb_hip_malloc": (_vp, [C.c_size_t]),
    "lsb_hip_free": (None, [_vp]),
    "lsb_hip_memcpy_h2d": (_i, [_vp, _vp, C.c_size_t]),
    "lsb_hip_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t]),
    "lsb_hip_sync": (_i, []),
}

_LIB = None


def build(force=False):
    """make -C lsbench_amd/csrc (hipcc --offload-arch=gfx950 + gcc)."""
    if force:
        subprocess.run(["make", "-s", "-C", _CSRC, "clean"], check=True)
    subprocess.run(["make", "-s", "-C", _CSRC, "all"], check=True)


def libc_free(ptr):
    """free() of a block a library function malloc'ed for the caller (lsb_csr_pad_lines' row map)."""
    libc = C.CDLL(None)
    libc.free.argtypes, libc.free.restype = [C.c_void_p], None
    libc.free(C.cast(ptr, C.c_void_p))


def load():
    """dlopen the library and attach the C signatures.  torch (when present) is
    imported first so that its bundled libamdhip64/librccl are the ones this
    library binds to -- two HIP runtimes in one process do not mix."""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a dependency
        pass
    if not os.path.exists(LIB_PATH):
        raise LsbenchHipError(
            "%s not built: run `make -C lsbench_amd/csrc` or __graft_entry__.build(); "
            "there is no fallback path" % LIB_PATH)
    hip = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    core = C.CDLL(CORE_PATH, mode=C.RTLD_GLOBAL)

    class _Both:
        """the two libraries behind one namespace: a symbol of include/lsbench.h
        comes from the stand-in core, everything else from the backend"""
        pass
    lib = _Both()
    lib.hip, lib.core = hip, core
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(core if name in CORE_SYMBOLS else hip, name)  # AttributeError = header/library mismatch
        fn.restype, fn.argtypes = res, args
        setattr(lib, name, fn)
    _LIB = lib
    return lib


def check(rc, what):
    if rc == 1:
        raise LsbenchHipError("%s: hip_cdna4 backend not initialised (no GPU, or "
                              "hip_cdna4_init not called) -- there is no CPU path" % what)
    if rc != 0:
        raise LsbenchHipError("%s: bad argument (rc=%d)" % (what, rc))
