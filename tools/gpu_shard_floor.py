"""What one GPU's share of config 3 costs per iteration WITHOUT communication:
the operator cut to 1/2, 1/4, 1/8 of its rows, single-reduction PCG (the
multi-GPU form) and classic PCG, fixed 2000 iterations."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import lsbench_amd as hip

hip.hip_cdna4_init()
for parts in (1, 2, 4, 8):
    ny = 3162 // parts
    A = hip.lsbench_matrix_synth(f"lap2d:nx=3162,ny={ny}")
    b = np.arange(A.nrows, dtype=np.float64)
    for kr in (hip.KRYLOV_PCG, hip.KRYLOV_PCG1):
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=1e-30, maxit=2000, krylov=kr, use_graph=0))
        s.solve(b)
        x, r = s.solve(b)
        print(f"1/{parts} of config 3 ({A.nrows} rows) krylov={kr} variant={s.spmv_variant} flags={s.spmv_flags}: "
              f"{r.seconds / r.iters * 1e6:.1f} us/iter", flush=True)
        s.destroy()
