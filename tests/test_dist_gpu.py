"""The multi-PROCESS path on ONE GPU.  RCCL refuses two ranks on a device, so
the ranks talk through tests/fake_rccl (a shared-memory test double preloaded in
front of librccl): everything of the product runs for real -- per-rank shard
generation, lsb_hip_solver_create_dist, the exchange plan, hip_comm.c's call
sequence, the identical control flow on every rank, the HIP kernels -- only the
transport under the nccl* symbols is replaced.  A mismatch in the sequence of
collectives between ranks shows up as a barrier time-out there, not as a hung
node in the 8-GPU bench."""
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest

import lsbench_amd as la
from conftest import ROOT
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FAKE = os.path.join(ROOT, "tests", "fake_rccl")


def _env():
    subprocess.run(["make", "-s", "-C", FAKE], check=True)
    env = dict(os.environ)
    env["LD_PRELOAD"] = os.path.join(FAKE, "libfake_rccl.so")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _run(world, args, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return r


@pytest.mark.parametrize("world,spec,krylov,comm,overlap", [
    (2, "lap2d:nx=240,ny=180", "cg", "rccl", 0),
    (3, "lap3d:nx=30,ny=28,nz=26", "cg1", "rccl", 0),
    (4, "lap2d:nx=150,ny=400", "auto", "rccl", 1),
    # the direct path: mailboxes opened through HIP IPC between the processes
    (2, "lap2d:nx=240,ny=180", "cg", "p2p", 0),
    (3, "lap3d:nx=30,ny=28,nz=26", "cg1", "p2p", 0),
    (4, "lap2d:nx=150,ny=400", "auto", "p2p", 1),
    (2, "lap2d:nx=300,ny=300", "auto", "auto", 0)])
def test_multi_process_solve_matches_oracle(world, spec, krylov, comm, overlap, tmp_path):
    import lsbench_amd as la
    _run(world, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), spec, str(tmp_path), krylov, "1e-10",
                 comm, str(overlap)], 29600 + world)
    A = la.lsbench_matrix_synth(spec)
    b = O.rhs(A.nrows)
    xo, ito, relo, sto = O.pcg_jacobi(A.offs, A.cols, A.vals, b, 1e-10)
    metas = [np.load(tmp_path / ("m%d.npy" % r)) for r in range(world)]
    x = np.concatenate([np.load(tmp_path / ("x%d.npy" % r)) for r in range(world)])
    y = np.concatenate([np.load(tmp_path / ("y%d.npy" % r)) for r in range(world)])
    assert [int(m[4]) for m in metas] == sorted(int(m[4]) for m in metas) and int(metas[-1][5]) == A.nrows
    assert all(m[1] == 1 and m[3] == 1 for m in metas)                  # converged, twice
    assert len({int(m[6]) for m in metas}) == 1                         # every rank took the same path
    if comm != "auto":
        assert int(metas[0][6]) == {"rccl": 1, "p2p": 3}[comm]
    assert all(int(m[7]) == overlap for m in metas)
    assert len({int(m[0]) for m in metas}) == 1 and len({int(m[2]) for m in metas}) == 1
    assert abs(int(metas[0][0]) - ito) <= 2 and int(metas[0][2]) == int(metas[0][0])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8
    v = np.sin(np.arange(A.nrows, dtype=np.float64))
    assert np.allclose(y, O.spmv(A.offs, A.cols, A.vals, v), rtol=1e-12, atol=1e-12)


def test_multi_process_gmres(tmp_path):
    """GMRES(m) across three processes: wide all-reduces (the j+1 Gram-Schmidt
    coefficients of a step in one call) through the communicator."""
    import lsbench_amd as la
    spec = "lap2d:nx=90,ny=70"
    _run(3, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), spec, str(tmp_path), "gmres", "1e-10",
             "rccl", "0"], 29611)
    A = la.lsbench_matrix_synth(spec)
    xo, _, _, _ = O.pcg_jacobi(A.offs, A.cols, A.vals, O.rhs(A.nrows), 1e-12)
    metas = [np.load(tmp_path / ("m%d.npy" % r)) for r in range(3)]
    x = np.concatenate([np.load(tmp_path / ("x%d.npy" % r)) for r in range(3)])
    assert all(m[1] == 1 and m[3] == 1 for m in metas)
    assert len({int(m[0]) for m in metas}) == 1 and int(metas[0][2]) == int(metas[0][0])
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 branch end to end (gloo for torch.distributed, the test
    double for the library's collectives): same iteration count as N = 1."""
    common = ["--workload", "lap2d:nx=700,ny=500", "--tol", "1e-8", "--steps", "2", "--warmup", "1",
              "--cpu-seconds", "0", "--cfg4", "0"]
    env = _env()
    env["LSB_BENCH_BACKEND"] = "gloo"
    env["LSB_BENCH_ONE_GPU"] = "1"
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29655", os.path.join(ROOT, "bench.py"),
           "--gpus", "2"] + common
    r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, (r2.stdout[-1500:], r2.stderr[-3000:])
    lines = [l for l in r2.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                              # rank 0 only
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["partition"] == "row-range x2"
    assert abs(two["config"]["iterations_per_solve"] - one["config"]["iterations_per_solve"]) <= 2
    assert two["config"]["relres"] <= 1e-8 and "single-reduction" in two["config"]["solver"]
    assert two["config"]["nnz"] == one["config"]["nnz"]
    # load balance of a sharded line: per-rank time per iteration and SpMV launch (min / max), rows, non-zeros
    pr = two["per_rank"]
    assert one["per_rank"] is None and len(pr["rows"]) == 2 and sum(pr["rows"]) == 350000
    assert sum(pr["nnz"]) == two["config"]["nnz"] and 0 < pr["iteration_us"]["min"] <= pr["iteration_us"]["max"]
    assert 0 <= pr["spmv_us"]["min"] <= pr["spmv_us"]["max"]


@pytest.mark.parametrize("ngpus,comm,matrix", [
    (2, "rccl", "xn3b_A_18"), (2, "p2p", "tj7a_A_18"), (3, "auto", "xn3b_A_12"),
    (4, "rccl", "synth:lap3d:nx=40,ny=36,nz=30")])
def test_driver_ngpus_from_one_process(ngpus, comm, matrix, matrix_path, golden_x):
    """`driver --solver hip --ngpus N`: the row-partitioned solve BEHIND the
    reference's backend contract -- one caller process, one hip_cdna4_bench call
    (src/lsbench-impl.h:42-68, bin/driver.c:5-15); the backend starts one host
    thread per GPU (hip_multi.c).  Rehearsed on the one GPU of this box: every
    rank on the caller's device (LSBENCH_HIP_SHARE_DEVICE=1), RCCL's transport
    under the nccl* symbols replaced by the test double, everything else --
    operator build, partition, per-rank shard, exchange plan, collectives'
    sequence, mailboxes reached through peer pointers, kernels, D2H of every
    rank's rows into the caller's x -- the product's own code."""
    drv = os.path.join(ROOT, "lsbench_amd", "csrc", "driver")
    env = _env()
    env.update(LSBENCH_HIP_SHARE_DEVICE="1", LSBENCH_HIP_COMM=comm, GPU_MAX_HW_QUEUES="8",
               LSBENCH_HIP_P2P_TIMEOUT_MS="20000")
    synth = matrix.startswith("synth:")
    path = matrix if synth else matrix_path(matrix)
    extra = ["--operator", "raw", "--tol", "1e-10"] if synth else []
    outs = {}
    for n in (1, ngpus):
        r = subprocess.run([drv, "--solver", "hip", "--matrix", path, "--trials=2", "--verbose", "2",
                            "--ngpus", str(n)] + extra, capture_output=True, text=True, env=env,
                           timeout=600)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
        lines = r.stdout.splitlines()
        f = lines[lines.index("===matrix,n,nnz,trials,solver,ordering,elapsed===") + 1].split(",")
        h = lines[lines.index("===hip_cdna4:iterations,relres,status,tol,solves_per_sec,nshards===") + 1].split(",")
        x = np.array([float(l.split("=")[1]) for l in lines if l.startswith("x[")])
        assert int(f[-4]) == 2 and int(f[-3]) == 6 and len(x) == int(f[-6])  # (a synth: name has commas)
        assert int(h[2]) == 1 and int(h[5]) == n                   # converged; N shards
        if n > 1:
            k = [i for i, l in enumerate(lines) if l.startswith("===hip_cdna4:ngpus,comm,rccl_ranks,")][0]
            got = lines[k + 1].split(",")
            assert int(got[0]) == n
            # the ranks the communicator itself counts (ncclCommCount), and rank 0's plan: a
            # banded operator exchanges halos with its one neighbour, not an all-gather
            assert int(got[2]) == n
            if synth:
                assert got[3] == "halos" and int(got[4]) == 1 and int(got[5]) == 1
                assert int(got[6]) == int(got[7]) == 40 * 36 * 8       # one plane of the 40 x 36 x 30 grid
            if comm == "rccl":
                assert got[1] == "rccl"
            if comm == "p2p":
                assert got[1].startswith("direct-xgmi")
        outs[n] = (x, int(h[0]))
    x1, it1 = outs[1]
    xn, itn = outs[ngpus]
    assert abs(it1 - itn) <= max(3, it1 // 50)                      # cg vs single-reduction cg
    if synth:
        A = la.lsbench_matrix_synth(matrix[6:])
        xo, _, _, _ = O.pcg_jacobi(A.offs, A.cols, A.vals, O.rhs(A.nrows), 1e-12)
        assert np.linalg.norm(xn - xo) / np.linalg.norm(xo) <= 1e-8
    else:
        xg = golden_x(matrix)
        assert np.linalg.norm(xn - xg) / np.linalg.norm(xg) <= 1e-10
    assert np.linalg.norm(xn - x1) / np.linalg.norm(x1) <= 1e-8


def test_hung_collective_ends_with_a_message():
    """A rank that never shows up must not hang the node: the host-side deadline of
    the sharded solve (opts.comm_deadline_s) turns it into exit code 1.  Here: two
    ranks are announced, one of them never joins the all-reduce because the test
    double is told to stall it."""
    drv = os.path.join(ROOT, "lsbench_amd", "csrc", "driver")
    env = _env()
    env.update(LSBENCH_HIP_SHARE_DEVICE="1", LSBENCH_HIP_COMM="rccl", FAKE_RCCL_STALL_RANK="1",
               FAKE_RCCL_STALL_AFTER="40", LSBENCH_HIP_COMM_DEADLINE_S="5")
    t0 = time.time()
    r = subprocess.run([drv, "--solver", "hip", "--matrix", "synth:lap2d:nx=300,ny=200", "--operator", "raw",
                        "--tol", "1e-10", "--trials=1", "--ngpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0
    assert "a collective of the sharded solve is hung" in r.stderr     # the product's deadline, not the double's
    # the stalled stream keeps "running" for 45 s (the double's host function sleeps whatever the
    # process does): the product has to leave without waiting for it -- _exit(), not exit()
    assert dt < 35, "the give-up path waited for a stream that cannot drain (%.0f s)" % dt
