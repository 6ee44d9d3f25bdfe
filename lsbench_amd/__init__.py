"""lsbench_amd -- MI355X-native sparse-solve backend for lsbench.

The product is the C/HIP library under lsbench_amd/csrc (C-ABI in include/);
this package is the ctypes binding the tests and bench.py drive it through.
"""
from . import _lib
from ._lib import (PRECOND_FSAI, PREC_FP64, PREC_MIXED, PRECOND_CHEBYSHEV, PRECOND_BLOCKJACOBI, SPMV_BINNED, SPMV_TWOPHASE, COMM_AUTO, COMM_P2P, COMM_RCCL, STATUS_COMM, KRYLOV_AUTO, KRYLOV_GMRES, KRYLOV_PCG, KRYLOV_PCG1, LsbenchHipError, OP_CHOLMOD_UPPER, OP_RAW, PRECOND_JACOBI, PRECOND_L1JACOBI,
                   PRECOND_NONE, SELL_ROWS, SPMV_FLAG_C16, SPMV_FLAG_TMPL, SPMV_FLAG_DEFER, SPMV_FLAG_COL, SPMV_FLAG_NT, SPMV_FLAG_PREFETCH, SPMV_ADAPTIVE, SPMV_AUTO, SPMV_PANEL, SPMV_SCALAR, SPMV_SELL,
                   SPMV_SUBWAVE, STATUS_BREAKDOWN, STATUS_CONVERGED,
                   STATUS_MAXIT, STATUS_RUNNING)
from .api import (Matrix, Solver, default_opts, hip_cdna4_bench,
                  hip_cdna4_finalize, hip_cdna4_init, last_result,
                  lsb_csr_bandwidth, lsb_csr_block_lanes, lsb_csr_col_hull,
                  lsb_csr_permute_sym, lsb_csr_rcm, lsb_csr_sellize, lsb_csr_sellize16, lsb_csr_copy_base0, lsb_csr_partition_rows,
                  lsb_plan_exchange,
                  lsb_csr_row_blocks, lsb_csr_row_slice,
                  lsb_csr_symmetrize_upper, lsbench_matrix_read,
                  lsbench_matrix_synth)

__all__ = [n for n in dir() if not n.startswith("_")]
