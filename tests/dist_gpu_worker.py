"""One rank of a multi-PROCESS solve on a single GPU (tests/test_dist_gpu.py):
launched by torch.distributed.run with LD_PRELOAD=tests/fake_rccl/libfake_rccl.so,
so that hip_comm.c's RCCL calls run between processes sharing cuda:0."""
import ctypes
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lsbench_amd as la  # noqa: E402


def main():
    spec, outdir, krylov, tol = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    comm = {"auto": la.COMM_AUTO, "rccl": la.COMM_RCCL, "p2p": la.COMM_P2P}[
        sys.argv[5] if len(sys.argv) > 5 else "auto"]
    overlap = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = la._lib.load()
    assert la.hip_cdna4_init() == 0
    idb = ctypes.create_string_buffer(la._lib.UNIQUE_ID_BYTES)
    if rank == 0:
        lib.lsb_hip_comm_get_unique_id(idb)
    t = torch.frombuffer(bytearray(idb.raw), dtype=torch.uint8).clone()
    dist.broadcast(t, 0)
    idb = ctypes.create_string_buffer(bytes(t.tolist()), la._lib.UNIQUE_ID_BYTES)
    la._lib.check(lib.lsb_hip_comm_init_rank(idb, world, rank), "comm_init_rank")
    n = la.lsbench_matrix_synth(spec, 0, 1).n_global
    r0 = (n * rank // world) & ~1
    r1 = (n * (rank + 1) // world) & ~1 if rank + 1 < world else n
    A = la.lsbench_matrix_synth(spec, r0, r1)
    kry = {"cg": la.KRYLOV_PCG, "cg1": la.KRYLOV_PCG1, "auto": la.KRYLOV_AUTO,
           "gmres": la.KRYLOV_GMRES}[krylov]
    s = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, tol=tol, krylov=kry, maxit=50000, comm=comm,
                                     overlap=overlap, spmv_variant=la.SPMV_ADAPTIVE if overlap else 0),
                  row_begin=r0, n_global=n)
    mode, p2p_us, rccl_us = s.comm
    d_b = torch.arange(r0, r1, dtype=torch.float64, device="cuda")
    d_x = torch.zeros(r1 - r0, dtype=torch.float64, device="cuda")
    res = s.solve_dev(d_b, d_x)
    res2 = s.solve_dev(d_b, d_x)  # second solve: iteration-count hint path
    # SpMV through the exchange as well
    d_v = torch.sin(torch.arange(r0, r1, dtype=torch.float64, device="cuda"))
    d_y = torch.empty_like(d_v)
    # many in a row, nothing in between (regression, round-1 advisor: with the
    # direct path a fast rank used to overwrite a halo region its peer was still
    # copying); every call on another vector, the last one is checked
    for k in range(30, -1, -1):
        s.spmv_dev(d_v * float(k + 1) if k else d_v, d_y)
    np.save(os.path.join(outdir, "x%d.npy" % rank), d_x.cpu().numpy())
    np.save(os.path.join(outdir, "y%d.npy" % rank), d_y.cpu().numpy())
    np.save(os.path.join(outdir, "m%d.npy" % rank),
            np.array([res.iters, res.status, res2.iters, res2.status, r0, r1, mode,
                      int(s.overlaps)], dtype=np.int64))
    if rank == 0:
        print("comm mode %d: direct %.1f us, rccl double %.1f us per exchange+all-reduce"
              % (mode, p2p_us, rccl_us), flush=True)
    s.destroy()
    lib.lsb_hip_comm_destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
