"""Run-to-run repeatability of the template SpMV kernels at full size (round 3: a sibling kernel,
since removed, was NOT repeatable on a 7-point grid; these are the ones that ship): config 4
(k_spmv_tmpl<2>, masks in 32 % of the slices), config 3 (<1>), a 1-D operator (<0>), 8 virtual
shards of config 4 (split launches): REPS SpMVs of the same vector and 3 solves each, every
result compared bit for bit with the first.  usage: gpu_repeat_stress.py [REPS]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
import lsbench_amd as la
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
assert la.hip_cdna4_init() == 0
bad = 0
for spec, kw, its in (("lap3d:nx=400,ny=400,nz=400", {}, 60), ("lap2d:nx=3162,ny=3162", {}, 400),
                      ("lap2d:nx=9000001,ny=1", {}, 30), ("lap3d:nx=400,ny=400,nz=400", dict(nvirt=8, overlap=1), 40),
                      ("lap3d:nx=400,ny=400,nz=50", {}, 100)):
    A = la.lsbench_matrix_synth(spec)
    n = A.nrows
    s = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, tol=0.0, maxit=its, verify=0, **kw))
    d_x = torch.sin(torch.arange(n, dtype=torch.float64, device="cuda") * 0.37)
    d_y0 = torch.empty(n, dtype=torch.float64, device="cuda")
    s.spmv_dev(d_x, d_y0)
    diff = 0
    d_y = torch.empty_like(d_y0)
    for k in range(reps):
        d_y.fill_(float("nan"))
        s.spmv_dev(d_x, d_y)
        diff += int((d_y != d_y0).sum().item())
    d_b = torch.arange(n, dtype=torch.float64, device="cuda")
    xs = []
    for k in range(3):
        d_s = torch.zeros(n, dtype=torch.float64, device="cuda")
        r = s.solve_dev(d_b, d_s)
        xs.append((d_s.clone(), int(r.iters), r.relres))
    same = all(torch.equal(xs[0][0], v[0]) and xs[0][1:] == v[1:] for v in xs[1:])
    print(f"{spec} {kw}: flags {s.spmv_flags} period {s.spmv_period} nt {s.blas1_nt}: {reps} SpMVs, {diff} differing elements; "
          f"3 x {its} iterations {'identical' if same else 'DIFFER'} (relres {xs[0][2]:.6e})", flush=True)
    bad += diff + (0 if same else 1)
    s.destroy()
    del d_x, d_y, d_y0, d_b, xs
    torch.cuda.empty_cache()
print("REPEATABLE" if bad == 0 else "NOT REPEATABLE")
sys.exit(1 if bad else 0)
