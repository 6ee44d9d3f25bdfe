"""first vs second (hinted) solve, template kernel vs per-slot kernel (diagnostic)"""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import lsbench_amd as hip
from oracle import oracle as O
hip.hip_cdna4_init()
spec = sys.argv[1] if len(sys.argv) > 1 else "lap3d:nx=64,ny=64,nz=40"
A = hip.lsbench_matrix_synth(spec)
b = O.rhs(A.nrows)
xo, ito, relo, sto = O.pcg_jacobi(A.offs, A.cols, A.vals, b, 1e-10)
for tune in (6, 70):
    for masks in ("0", "1"):
        if masks == "0":
            os.environ["LSBENCH_HIP_NO_MASKS"] = "1"
        else:
            os.environ.pop("LSBENCH_HIP_NO_MASKS", None)
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, spmv_variant=hip.SPMV_SELL, tol=1e-10, spmv_tune=tune, use_graph=0))
        for k in range(3):
            x, r = s.solve(b)
            print(spec, "tune", tune, "masks", masks, "solve", k, "iters", r.iters, "relres %.3e" % r.relres, "status", r.status,
                  "|x-xo|/|xo| %.2e" % (np.linalg.norm(x - xo) / np.linalg.norm(xo)), "fused", s.fused_p, flush=True)
        s.destroy()
print("oracle iters", ito)
