#!/usr/bin/env python3
"""What RCM buys the SpMV: a 4M-row 5-point Laplacian with randomly shuffled
rows/columns (no locality left) vs the same operator after reorder=1."""
import os, sys, time
import numpy as np, scipy.sparse as sp, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lsbench_amd as la
assert la.hip_cdna4_init() == 0
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
A = la.lsbench_matrix_synth("lap2d:nx=%d,ny=%d" % (nx, nx)); n = A.nrows
M = sp.csr_matrix((A.vals, A.cols.astype(np.int64), A.offs.astype(np.int64)), shape=(n, n))
q = np.random.default_rng(0).permutation(n); Ms = M[q][:, q].tocsr(); Ms.sort_indices()
S = la.Matrix.from_arrays(Ms.indptr, Ms.indices, Ms.data)
B = 12 * S.nnz + 20 * n + 4
for name, mat, reorder in [("banded (generator order)", A, 0), ("shuffled", S, 0), ("shuffled + RCM", S, 1)]:
    t = time.time(); s = la.Solver(mat, la.default_opts(op_mode=la.OP_RAW, reorder=reorder)); ts = time.time() - t
    ms = s.time_spmv(3, 20)
    print("%-26s setup %.2fs  spmv %.1f us = %.0f GB/s (%.1f%% of 8 TB/s)" % (name, ts, ms * 1e3, B / ms / 1e6, B / ms / 1e6 / 80))
    s.destroy()
