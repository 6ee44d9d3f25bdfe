"""Config 4 (7-point 400^3, 64 M rows): what ONE rank's share -- 50 planes, 8 M rows, two
1.28 MB halos -- costs per iteration of the sharded solve, measured on one GPU (VERDICT r2
item 5), against the single-GPU iteration of the whole operator:
  whole    config 4 on one shard, classic PCG (what `cfg4` of the N = 1 bench line runs)
  alone    nz = 50 slab as the only rank of a communicator (LSBENCH_HIP_DIST_ALONE=1):
           single-reduction CG with its exchange / all-reduce launches in place, nobody to wait
           for, no halo (a single rank owns every column)
  virt8    the whole operator as 8 row-range shards on the one device, per-iteration time / 8:
           each shard's kernels + its halo copies (device copies or the direct path's mailbox
           kernels), interior / boundary split of the SpMV with overlap on
The implied ceiling of the 8-GPU run is whole / share -- communication latency over xGMI comes
on top of `share` there.  usage: gpu_dist_floor4.py [iterations]"""
import ctypes, os, sys
sys.path.insert(0, ".")
import numpy as np
import torch
import lsbench_amd as la

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
lib = la._lib.load()
assert la.hip_cdna4_init() == 0


def run(A, name, **kw):
    n = A.nrows
    rb = kw.pop("row_begin", None)
    o = la.default_opts(op_mode=la.OP_RAW, tol=1e-30, maxit=iters, verify=0, **kw)
    s = la.Solver(A, o, row_begin=rb, n_global=n) if rb is not None else la.Solver(A, o)
    d_b = torch.arange(n, dtype=torch.float64, device="cuda")
    d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
    s.solve_dev(d_b, d_x)
    r = s.solve_dev(d_b, d_x)
    us = r.seconds / r.iters * 1e6
    print(f"{name:34s} variant={s.spmv_variant} flags={s.spmv_flags} period={s.spmv_period} comm={s.comm[0]} "
          f"overlap={int(s.overlaps)}: {us:8.1f} us/iter ({r.iters} iterations)", flush=True)
    s.destroy()
    del d_b, d_x
    torch.cuda.empty_cache()
    return us


A = la.lsbench_matrix_synth("lap3d:nx=400,ny=400,nz=400")
whole = run(A, "whole, one shard (classic PCG)")
whole1 = run(A, "whole, one shard (single-reduction)", krylov=la.KRYLOV_PCG1)
res = {}
for comm, cname in ((la.COMM_RCCL, "device copies"), (la.COMM_P2P, "direct path")):
    for ov in (0, 1):
        us = run(A, f"virt8 {cname} overlap={ov}", nvirt=8, comm=comm, overlap=ov, krylov=la.KRYLOV_AUTO)
        res[(cname, ov)] = us / 8
        print(f"    -> per shard {us / 8:8.1f} us/iter; ceiling {whole / (us / 8):.2f}x of the one-GPU iteration")
    # round 4: opts.overlap = -1 -- both forms timed on this communicator at creation, the faster one runs
    n = A.nrows
    s = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, tol=1e-30, maxit=iters, verify=0, nvirt=8, comm=comm,
                                     krylov=la.KRYLOV_AUTO))
    print(f"virt8 {cname} overlap=-1 (timed at creation): {s.comm_plan['overlap_timed_us']} -> "
          f"{'split' if s.overlaps else 'plain'}", flush=True)
    s.destroy()
del A
os.environ["LSBENCH_HIP_DIST_ALONE"] = "1"
idb = ctypes.create_string_buffer(la._lib.UNIQUE_ID_BYTES)
lib.lsb_hip_comm_get_unique_id(idb)
la._lib.check(lib.lsb_hip_comm_init_rank(idb, 1, 0), "comm_init_rank")
S = la.lsbench_matrix_synth("lap3d:nx=400,ny=400,nz=50")
for comm, cname in ((la.COMM_RCCL, "rccl"), (la.COMM_P2P, "direct")):
    us = run(S, f"alone (nz = 50 slab) {cname}", comm=comm, krylov=la.KRYLOV_PCG1, row_begin=0)
    print(f"    -> ceiling {whole / us:.2f}x of the one-GPU iteration")
lib.lsb_hip_comm_destroy()
