"""Config 3 to convergence through the multi-GPU code path on one device: 8
row-range shards, single-reduction PCG (implicit u), both transports; iteration
count and recomputed residual against the one-shard classic solve."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import lsbench_amd as hip

hip.hip_cdna4_init()
A = hip.lsbench_matrix_synth("lap2d:nx=3162,ny=3162")
n = A.nrows
d_b = torch.arange(n, dtype=torch.float64, device="cuda")
for nv, comm, kr in ((1, hip.COMM_AUTO, hip.KRYLOV_PCG), (1, hip.COMM_AUTO, hip.KRYLOV_PCG1),
                     (8, hip.COMM_RCCL, hip.KRYLOV_AUTO), (8, hip.COMM_P2P, hip.KRYLOV_AUTO)):
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, nvirt=nv, comm=comm, krylov=kr, tol=1e-8,
                                       maxit=100000, use_graph=0))
    d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
    t = time.perf_counter()
    r = s.solve_dev(d_b, d_x)
    dt = time.perf_counter() - t
    d_y = torch.empty_like(d_x)
    s.spmv_dev(d_x, d_y)
    true = float(((d_b - d_y) ** 2).sum().sqrt() / (d_b ** 2).sum().sqrt())
    print(f"nvirt={nv} comm={s.comm[0]} krylov={kr}: iters={r.iters} status={r.status} relres={r.relres:.3e} "
          f"true={true:.3e} {dt / r.iters * 1e6:.1f} us/iter", flush=True)
    s.destroy()
