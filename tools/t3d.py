"""Sliced-ELL timing pass on a 3-D stencil, verbose: what the plane-periodic XCD
dealing (sell_deal, hip_kernels.hip) does to the launch time.  python tools/t3d.py [spec]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lsbench_amd as la
from oracle import oracle as O
la.hip_cdna4_init()
spec = sys.argv[1] if len(sys.argv) > 1 else "lap3d:nx=128,ny=128,nz=40"
A = la.lsbench_matrix_synth(spec)
x = np.random.default_rng(0).standard_normal(A.nrows)
yo = O.spmv(A.offs, A.cols, A.vals, x, threads=16) if A.nrows < 70000000 else None
s = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, spmv_variant=la.SPMV_SELL, verbose=2))
d_y = torch.empty(A.nrows, dtype=torch.float64, device="cuda")
s.spmv_dev(torch.from_numpy(x).cuda(), d_y)
print("picked: flags", s.spmv_flags, "grid", s.spmv_grid, "period", s.spmv_period,
      "maxerr", np.abs(d_y.cpu().numpy() - yo).max(), "ms", s.time_spmv(5, 50))
s.destroy()
