"""The same question as gpu_placement_probe.py for config 4 (512 MB vectors: nothing fits a cache):
does the iteration time depend on where the vectors land?  LSBENCH_HIP_NO_PLACEMENT=1 is set."""
import os, sys
os.environ["LSBENCH_HIP_NO_PLACEMENT"] = "1"
os.environ["LSBENCH_HIP_BLAS1_NT"] = "41"
sys.path.insert(0, ".")
import torch
import lsbench_amd as la

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
assert la.hip_cdna4_init() == 0
A = la.lsbench_matrix_synth("lap3d:nx=400,ny=400,nz=400")
n = A.nrows
keep = []
for k in range(rounds):
    if k:
        keep.append(torch.empty((k * 37 + 11) * (1 << 20), dtype=torch.uint8, device="cuda"))
    o = la.default_opts(op_mode=la.OP_RAW, tol=1e-30, maxit=100, verify=0, sample_spmv=16)
    s = la.Solver(A, o)
    d_b = torch.arange(n, dtype=torch.float64, device="cuda")
    d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
    s.solve_dev(d_b, d_x)
    res = []
    for rep in range(2):
        r = s.solve_dev(d_b, d_x)
        res.append((r.seconds / r.iters * 1e6, r.spmv_ms * 1e3))
    print("solver %d: flags=%d period=%d  " % (k, s.spmv_flags, s.spmv_period) + "  ".join("%.1f us/iter (SpMV %.1f)" % t for t in res), flush=True)
    s.destroy()
    del d_b, d_x
