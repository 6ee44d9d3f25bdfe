#!/usr/bin/env python3
"""Power-law (config 5) SpMV timings through the library: kernel variants and flags."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsbench_amd as la
assert la.hip_cdna4_init() == 0
G = 1.585350372615855
for n in [int(a) for a in sys.argv[1:]] or [2000000, 8000000]:
    A = la.lsbench_matrix_synth("powerlaw:n=%d,gamma=%r,max=4096,seed=20240607" % (n, G))
    B = 12 * A.nnz + 20 * A.nrows + 4
    lens = np.diff(A.offs.astype(np.int64))
    print("n=%d nnz=%d mean=%.1f max=%d rows>2048: %d (%.1f%% of nnz) rows>256: %d (%.1f%% of nnz)" % (
        n, A.nnz, lens.mean(), lens.max(), (lens > 2048).sum(), 100 * lens[lens > 2048].sum() / A.nnz,
        (lens > 256).sum(), 100 * lens[lens > 256].sum() / A.nnz))
    for variant, tune in [(1, 0), (1, 3), (2, 0), (4, 0), (4, 1), (4, 2), (4, 3), (0, -1)]:
        o = la.default_opts(op_mode=la.OP_RAW, precond=la.PRECOND_NONE, spmv_variant=variant, spmv_tune=tune)
        s = la.Solver(A, o)
        ms = s.time_spmv(3, 10)
        print("  variant=%d(->%d) flags=%d(->%d): %.1f us => %.0f GB/s (%.1f%%)" % (variant, s.spmv_variant, tune, s.spmv_flags, ms * 1e3, B / ms / 1e6, B / ms / 1e6 / 80))
        s.destroy()
