"""The N>1 host logic on the CPU: world_size-2 (and 3) gloo ranks run the
row-partitioned PCG exactly as hip_pcg.c / hip_dist.c sequence it (exchange -> SpMV on
the global-index vector -> all-reduce(p.q) -> x/r update -> all-reduce(r.z,r.r)
-> p update), with the PRODUCT's partitioner, shard generator and exchange plan
(C, through the C-ABI) and the oracle's CPU kernels standing in for the HIP
kernels (test infrastructure only -- the product has no CPU path).  The result
must reproduce the one-rank oracle solve."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_main(rank, world, port, spec, tol, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import lsbench_amd as la
    from oracle import oracle as O

    # every rank derives the same row bounds (even rows, like bench.py does)
    full = la.lsbench_matrix_synth(spec, 0, 1)  # 1 row, just to learn n_global
    n = full.n_global
    bounds = [(n * p // world) & ~1 for p in range(world)] + [n]
    r0, r1 = bounds[rank], bounds[rank + 1]
    A = la.lsbench_matrix_synth(spec, r0, r1)          # product generator, own rows only
    lo, hi = la.lsb_csr_col_hull(A)
    mine = torch.tensor([r0, r1 - r0, lo, hi], dtype=torch.int64)
    allh = [torch.zeros(4, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allh, mine)
    hull = torch.stack(allh).numpy()
    recvs, sends = la.lsb_plan_exchange(rank, hull)    # product plan (C)

    offs, cols, vals = A.offs.astype(np.uint64), A.cols.copy(), A.vals.copy()
    nl = r1 - r0
    diag = np.array([vals[offs[i]:offs[i + 1]][cols[offs[i]:offs[i + 1]] == r0 + i][0]
                     for i in range(nl)])
    dinv = 1.0 / diag
    b = np.arange(r0, r1, dtype=np.float64)            # b_i = i (src/lsbench.c:159-160)
    pfull = np.zeros(n)                                 # GLOBAL index space
    x = np.zeros(nl)
    r = b.copy()
    pfull[r0:r1] = dinv * b

    def allreduce(*v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t)
        return t.tolist()

    def exchange():
        reqs, bufs = [], []
        for peer, off, cnt in recvs:
            buf = torch.empty(cnt, dtype=torch.float64)
            bufs.append((off, cnt, buf))
            reqs.append(dist.irecv(buf, src=peer))
        for peer, off, cnt in sends:
            reqs.append(dist.isend(torch.from_numpy(pfull[off:off + cnt].copy()), dst=peer))
        for q in reqs:
            q.wait()
        for off, cnt, buf in bufs:
            pfull[off:off + cnt] = buf.numpy()

    rz, bb = allreduce(float(r @ pfull[r0:r1]), float(b @ b))
    it, status = 0, 3
    while it < 5000:
        exchange()
        q = O.spmv(offs, cols, vals, pfull)
        (pq,) = allreduce(float(pfull[r0:r1] @ q))
        alpha = rz / pq
        x += alpha * pfull[r0:r1]
        r -= alpha * q
        rz_new, rr = allreduce(float(r @ (dinv * r)), float(r @ r))
        it += 1
        if rr <= tol * tol * bb:
            status = 1
            break
        pfull[r0:r1] = dinv * r + (rz_new / rz) * pfull[r0:r1]
        rz = rz_new
    np.save(os.path.join(out_dir, "x%d.npy" % rank), x)
    np.save(os.path.join(out_dir, "meta%d.npy" % rank),
            np.array([it, status, len(recvs), len(sends), sum(c for _, _, c in recvs)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,spec", [(2, "lap2d:nx=37,ny=29"), (3, "lap3d:nx=9,ny=8,nz=11"),
                                        (2, "lap3d:nx=12,ny=10,nz=9")])
def test_row_partitioned_pcg_matches_single_rank(world, spec, tmp_path):
    from oracle import oracle as O
    import lsbench_amd as la
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(world, port, spec, 1e-10, str(tmp_path)), nprocs=world, join=True)
    x = np.concatenate([np.load(tmp_path / ("x%d.npy" % r)) for r in range(world)])
    metas = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(world)]
    A = la.lsbench_matrix_synth(spec)
    x1, it1, rel1, st1 = O.pcg_jacobi(A.offs, A.cols, A.vals, O.rhs(A.nrows), 1e-10)
    assert st1 == 1 and all(m[1] == 1 for m in metas)
    assert all(abs(int(m[0]) - it1) <= 1 for m in metas)          # same iteration count
    assert len({int(m[0]) for m in metas}) == 1                   # identical on every rank
    assert np.linalg.norm(x - x1) / np.linalg.norm(x1) < 1e-9
    # banded operator: neighbours only, one grid line/plane each -- not an all-gather
    for rk, m in enumerate(metas):
        assert m[2] == (1 if rk in (0, world - 1) else 2)
        assert m[4] < A.nrows // 2
