"""What ONE rank's share of config 3 costs per iteration WITH the communication
launches of a sharded solve in place (a communicator of one rank,
LSBENCH_HIP_DIST_ALONE=1: exchange, all-reduce and single-reduction CG run as
they would on 8 GPUs, only nobody has to be waited for):
  rccl         all-reduce = reduction launch + ncclAllReduce
  direct       all-reduce = k_p2p_allreduce, one launch          (LSBENCH_HIP_AR_FOLD=0)
  direct+collect  contribute = a launch that waits for nobody, wait + collect at the head
               of k_cg1_update                                    (LSBENCH_HIP_AR_FOLD=1, default)
  direct+fold  contribute in the SpMV's tail as well: no launch   (LSBENCH_HIP_AR_FOLD=2)
and that folding leaves every bit of x where it was.  VERDICT r1 item 8.
usage: gpu_dist_floor.py [iterations [shares, e.g. 8,4,1]]"""
import ctypes, os, sys
sys.path.insert(0, ".")
os.environ["LSBENCH_HIP_DIST_ALONE"] = "1"
import numpy as np
import torch
import lsbench_amd as la

lib = la._lib.load()
assert la.hip_cdna4_init() == 0
idb = ctypes.create_string_buffer(la._lib.UNIQUE_ID_BYTES)
lib.lsb_hip_comm_get_unique_id(idb)
la._lib.check(lib.lsb_hip_comm_init_rank(idb, 1, 0), "comm_init_rank")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
shares = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 4, 1]
for parts in shares:
    ny = 3162 // parts
    A = la.lsbench_matrix_synth(f"lap2d:nx=3162,ny={ny}")
    n = A.nrows
    d_b = torch.arange(n, dtype=torch.float64, device="cuda")
    ref = {}
    for name, comm, fold in (("rccl", la.COMM_RCCL, "0"), ("direct", la.COMM_P2P, "0"),
                             ("direct+collect", la.COMM_P2P, "1"), ("direct+fold", la.COMM_P2P, "2")):
        os.environ["LSBENCH_HIP_AR_FOLD"] = fold
        s = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, tol=1e-30, maxit=iters, comm=comm, verify=0,
                                         krylov=la.KRYLOV_PCG1),
                      row_begin=0, n_global=n)
        d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
        s.solve_dev(d_b, d_x)
        r = s.solve_dev(d_b, d_x)
        x = d_x.cpu().numpy()
        ref[name] = x
        same = ""
        if name in ("direct+collect", "direct+fold"):  # (the RCCL form reduces its partial sums in another order)
            same = " -- same bits as direct" if np.array_equal(x, ref["direct"]) else " -- DIFFERS from direct"
            same += ", |x - x_rccl|/|x_rccl| = %.1e" % (np.linalg.norm(x - ref["rccl"]) / np.linalg.norm(ref["rccl"]))
        print(f"1/{parts} of config 3 ({n} rows) {name:14s} variant={s.spmv_variant} comm mode {s.comm[0]}: "
              f"{r.seconds / r.iters * 1e6:.1f} us/iter ({r.iters} iterations){same}", flush=True)
        s.destroy()
lib.lsb_hip_comm_destroy()
