"""Does the per-iteration time of config 3 depend on WHERE the solver's vectors land?  One
process, several solvers created one after the other (each allocates its vectors anew; between
them a dummy allocation of a varying size shifts what the next one gets), 300 iterations each.
usage: gpu_placement_probe.py [rounds] [spec] [precond: jacobi|cheb|bj]"""
import sys
sys.path.insert(0, ".")
import torch
import lsbench_amd as la

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
assert la.hip_cdna4_init() == 0
spec = sys.argv[2] if len(sys.argv) > 2 else "lap2d:nx=3162,ny=3162"
prec = {"jacobi": la.PRECOND_JACOBI, "cheb": la.PRECOND_CHEBYSHEV, "bj": la.PRECOND_BLOCKJACOBI}[sys.argv[3] if len(sys.argv) > 3 else "jacobi"]
A = la.lsbench_matrix_synth(spec)
n = A.nrows
keep = []
for k in range(rounds):
    if k:
        keep.append(torch.empty((k * 37 + 11) * (1 << 20), dtype=torch.uint8, device="cuda"))  # shifts the heap
    o = la.default_opts(op_mode=la.OP_RAW, tol=1e-30, maxit=300 if prec == la.PRECOND_JACOBI else 60, verify=0, sample_spmv=16, precond=prec, cheb_degree=4)
    s = la.Solver(A, o)
    d_b = torch.arange(n, dtype=torch.float64, device="cuda")
    d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
    s.solve_dev(d_b, d_x)
    res = []
    for rep in range(3):
        r = s.solve_dev(d_b, d_x)
        res.append((r.seconds / r.iters * 1e6, r.spmv_ms * 1e3))
    print("solver %d: nt=%d  " % (k, s.blas1_nt) + "  ".join("%.1f us/iter (SpMV %.1f)" % t for t in res), flush=True)
    s.destroy()
    del d_b, d_x
