"""SpMV launch time back to back vs inside the solve (same solver object).
gpurun -- python tools/gpu_spmv_ctx.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import lsbench_amd as hip

hip.hip_cdna4_init()
A = hip.lsbench_matrix_synth("lap2d:nx=3162,ny=3162")
b = np.arange(A.nrows, dtype=np.float64)
for kr in (hip.KRYLOV_PCG, hip.KRYLOV_PCG1):
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=1e-30, maxit=800, krylov=kr,
                                       sample_spmv=16, use_graph=0))
    print("flags", s.spmv_flags, "grid", s.spmv_grid, flush=True)
    for rep in range(2):
        ms = s.time_spmv(20, 200)
        x, r = s.solve(b)
        print(f"krylov={kr} back-to-back {ms*1e3:.1f} us | in-solve {r.spmv_ms*1e3:.1f} us "
              f"({r.spmv_samples} samples) | iteration {r.seconds/r.iters*1e6:.1f} us", flush=True)
    s.destroy()
