"""Cost of the split (interior / boundary) SpMV launches: virtual shards on one
device, exchange-behind-interior on and off.  gpurun -- python tools/gpu_overlap.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import lsbench_amd as hip

hip.hip_cdna4_init()
spec = sys.argv[1] if len(sys.argv) > 1 else "lap2d:nx=3162,ny=3162"
A = hip.lsbench_matrix_synth(spec)
b = np.arange(A.nrows, dtype=np.float64)
for nv in (1, 2, 8):
    for kr in (hip.KRYLOV_PCG, hip.KRYLOV_PCG1):
        for ov, cm in ((0, 0), (1, 0), (0, 2), (1, 2)):
            if nv == 1 and (ov or cm):
                continue
            s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, nvirt=nv, overlap=ov, tol=1e-30,
                                               maxit=600, krylov=kr, comm=cm))
            s.solve(b)
            t = time.perf_counter()
            x, r = s.solve(b)
            dt = time.perf_counter() - t
            print(f"{spec} nvirt={nv} krylov={kr} overlap={ov} comm={s.comm} "
                  f"iters={r.iters} us/iter={dt / r.iters * 1e6:.1f}", flush=True)
            s.destroy()
