"""Config 4 (7-point 400^3): the template SpMV back to back and the classic PCG iteration under
the experiment switches of the environment this process was started with.
usage: gpu_cfg4_probe.py [label] [iterations] [spec]"""
import os, sys
sys.path.insert(0, ".")
import torch
import lsbench_amd as la

label = sys.argv[1] if len(sys.argv) > 1 else "base"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
spec = sys.argv[3] if len(sys.argv) > 3 else "lap3d:nx=400,ny=400,nz=400"
assert la.hip_cdna4_init() == 0
A = la.lsbench_matrix_synth(spec)
n = A.nrows
kw = {}
if os.environ.get("PROBE_KRYLOV") == "cg1":
    kw["krylov"] = la.KRYLOV_PCG1
if os.environ.get("PROBE_TUNE"):
    kw["spmv_variant"] = la.SPMV_SELL
    kw["spmv_tune"] = int(os.environ["PROBE_TUNE"])
if os.environ.get("PROBE_GRID"):
    kw["spmv_grid"] = int(os.environ["PROBE_GRID"])
o = la.default_opts(op_mode=la.OP_RAW, tol=1e-30, maxit=iters, verify=0,
                    sample_spmv=0 if os.environ.get("PROBE_NOSAMPLE") else 16, **kw)   # (sampling keeps the three-launch form)
s = la.Solver(A, o)
ms = s.time_spmv(10, 100)
d_b = torch.arange(n, dtype=torch.float64, device="cuda")
d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
d_y = torch.empty_like(d_x)
s.spmv_dev(torch.sin(d_b), d_y)
chk = int(d_y.view(torch.int64).sum())
s.solve_dev(d_b, d_x)
r = s.solve_dev(d_b, d_x)
print(f"{label:28s} flags={s.spmv_flags} grid={s.spmv_grid} period={s.spmv_period} nt={s.blas1_nt} fused_p={s.fused_p}: "
      f"SpMV back to back {ms * 1e3:7.1f} us; iteration {r.seconds / r.iters * 1e6:7.1f} us "
      f"(SpMV in the solve {r.spmv_ms * 1e3:7.1f} us) y checksum {chk & 0xffffffffffff:012x}", flush=True)
s.destroy()
