#!/usr/bin/env python3
"""First-contact check on a real MI355X: kernels vs oracle, PCG vs golden, raw
SpMV timings for every variant on the 10M-row Laplacian.  Development tool, not
a test (tests/ holds the parity suite)."""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsbench_amd as la  # noqa: E402
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def mat_path(name):
    p = os.path.join(GOLD, "matrices", name + ".txt")
    if os.path.exists(p):
        return p
    out = os.path.join(tempfile.gettempdir(), name + ".txt")
    if not os.path.exists(out):
        with gzip.open(p + ".gz", "rb") as fi, open(out, "wb") as fo:
            fo.write(fi.read())
    return out


def main():
    print("torch", torch.__version__, "hip", torch.version.hip, "dev", torch.cuda.get_device_name(0))
    rc = la.hip_cdna4_init()
    print("hip_cdna4_init rc", rc)
    assert rc == 0
    dev = "cuda:0"

    # ---- PCG vs golden on the reference matrices -------------------------
    meta = json.load(open(os.path.join(GOLD, "golden.json")))["matrices"]
    for name in ["I1_05x05", "A0_02x02", "A1_02x02", "xn3b_A_18", "tj7a_A_18"]:
        A = la.lsbench_matrix_read(mat_path(name))
        xg = np.fromfile(os.path.join(GOLD, "x", name + ".x.f64"), "<f8")
        for graph in (0, 1):
            for nv in (1, 3):
                o = la.default_opts(use_graph=graph, nvirt=nv)
                t = time.time()
                x = la.hip_cdna4_bench(A, trials=2, matrix_name=name, opts=o)
                dt = time.time() - t
                r = la.last_result()
                err = np.linalg.norm(x - xg) / max(np.linalg.norm(xg), 1e-300)
                print("%-10s graph=%d nvirt=%d iters=%d (oracle %d) status=%d relres=%.2e err_vs_golden=%.2e solve=%.3f ms wall=%.2fs" % (
                    name, graph, nv, r.iters, meta[name]["pcg_tol1e-12"]["iters"], r.status,
                    r.relres, err, r.seconds * 1e3, dt))
                assert err < 1e-10, err

    # ---- SpMV variants vs oracle ------------------------------------------
    lib = la._lib.load()
    for spec in ["lap2d:nx=300,ny=200", "lap3d:nx=30,ny=20,nz=25",
                 "powerlaw:n=30000,gamma=1.585350372615855,max=4096,seed=5"]:
        A = la.lsbench_matrix_synth(spec)
        n, nnz = A.nrows, A.nnz
        rng = np.random.default_rng(1)
        xh = rng.standard_normal(n)
        yo = O.spmv(A.offs, A.cols, A.vals, xh)
        absb = O.spmv(A.offs, A.cols, np.abs(A.vals), np.abs(xh))
        rb = la.lsb_csr_row_blocks(A, 2048)
        d_bl = torch.from_numpy(la.lsb_csr_block_lanes(A, rb)).to(dev)
        d_offs = torch.from_numpy(A.offs.astype(np.int32)).to(dev)
        d_cols = torch.from_numpy(A.cols.astype(np.int32)).to(dev)
        d_vals = torch.from_numpy(A.vals.copy()).to(dev)
        d_rb = torch.from_numpy(rb.astype(np.int32)).to(dev)
        d_x = torch.from_numpy(xh).to(dev)
        d_w = torch.zeros(lib.lsb_hip_partials_capacity(), dtype=torch.float64, device=dev)
        d_dot = torch.zeros(1, dtype=torch.float64, device=dev)
        for variant, mean in [(1, 0), (2, 2), (2, 4), (2, 8), (2, 16), (2, 32), (2, 64), (3, 0)]:
            d_y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
            rc = lib.lsb_hip_spmv_csr_f64(variant, n, d_offs.data_ptr(), d_cols.data_ptr(),
                                          d_vals.data_ptr(), d_rb.data_ptr(), d_bl.data_ptr(),
                                          len(rb) - 1, mean, 3, d_x.data_ptr(), d_y.data_ptr(), d_x.data_ptr(),
                                          d_dot.data_ptr(), d_w.data_ptr(), lib.lsb_hip_stream())
            assert rc == 0
            lib.lsb_hip_sync()
            y = d_y.cpu().numpy()
            viol = np.abs(y - yo) - 4 * np.finfo(float).eps * np.maximum(np.diff(A.offs), 1) * absb
            dot = d_dot.item()
            print("%-40s variant=%d L=%-2d max|dy|=%.2e viol=%.2e dot rel err=%.2e" % (
                spec[:40], variant, mean, np.abs(y - yo).max(), viol.max(),
                abs(dot - xh @ yo) / abs(xh @ yo)))
            assert viol.max() <= 0

    # ---- raw SpMV speed on the 10M-row Laplacian ----------------------------
    t = time.time()
    A = la.lsbench_matrix_synth("lap2d:nx=3162,ny=3162")
    print("gen lap2d 10M: %.2fs nnz=%d" % (time.time() - t, A.nnz))
    B = 12 * A.nnz + 20 * A.nrows + 4
    for variant in (1, 2, 3):
        o = la.default_opts(op_mode=la.OP_RAW, spmv_variant=variant, use_graph=0, maxit=50,
                            tol=0.0, sample_spmv=5)
        t = time.time()
        s = la.Solver(A, o)
        ts = time.time() - t
        ms = s.time_spmv(5, 50)
        d_b = torch.arange(A.nrows, dtype=torch.float64, device=dev)
        d_x = torch.zeros(A.nrows, dtype=torch.float64, device=dev)
        r = s.solve_dev(d_b, d_x)
        print("lap2d-10M variant=%d setup=%.2fs spmv=%.1f us => %.0f GB/s (%.1f%% of 8TB/s); "
              "50 PCG its: %.2f ms/iter, in-solve spmv %.1f us (%d samples) relres=%.3e" % (
                  variant, ts, ms * 1e3, B / ms / 1e6, B / ms / 1e6 / 80, r.seconds * 1e3 / 50,
                  r.spmv_ms * 1e3, r.spmv_samples, r.relres))
        s.destroy()
    print("SANITY OK")


if __name__ == "__main__":
    main()
