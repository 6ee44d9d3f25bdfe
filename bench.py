#!/usr/bin/env python3
"""Headline benchmark of the hip_cdna4 backend (BASELINE.json metric:
"fp64 CSR SpMV GB/s (% HBM roofline) + CG solves/sec at 1/2/4/8 MI355X").

One "step" = one Jacobi-PCG solve of the workload to the stated tolerance from
x0 = 0 with b_i = i (reference RHS, src/lsbench.c:157-160), operator and b
already resident in HBM.  value = solves / second, whole job.  For N > 1 the
same operator is row-range partitioned over the ranks (strong scaling): one
process per GPU, halo exchange + dot-product all-reduces over RCCL inside the
C library, launched as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Default workload = BASELINE.json configs[2], the roofline run: 5-point 2-D
Laplacian, 3162^2 = 9,998,244 rows, 49,978,572 nnz.  Every line also carries a
`cfg4` sub-record: configs[3] (7-point 400^3, 64 M rows, the configuration the
">= 6x at 8 GPUs" target is stated on) solved on the same N GPUs right after the
headline -- take the 1 -> 8 ratio of cfg4.value from the N = 1 and N = 8 lines.
At N = 1 the line also carries `general_values`: the SAME 3162^2 pattern with
general values (`lap2d:...,coef=1`, one hashed weight per grid edge), i.e. the
thing the metric names -- an fp64 CSR SpMV whose values must be streamed -- with
its roofline fraction on SURVEY 8(d)'s CSR byte count.  The constant-coefficient
headline stores ONE value per constant diagonal of a slice, so its `roofline`
is quoted on the bytes that layout must move (x once, y once, slot records, the
kept values), never on bytes it does not move; the CSR-count figure stays in
the record as `csr_count`.
Others:
    --workload lap3d     configs[3]: 7-point 400^3, 64 M rows (the 8-GPU config)
    --workload powerlaw  configs[4]: SpMV-only (unsymmetric), reports GB/s
    --workload file:tests/golden/matrices/xn3b_A_18.txt.gz   configs[1]

The JSON line also carries
  roofline      dominant kernel (SpMV): bytes the stored layout must move per launch
                / mean launch time, HIP events on the library's stream, sampled
                INSIDE the timed solves; peak 8000 GB/s
  cpu_baseline  the oracle's OpenMP Jacobi-PCG (a CPU port of the same
                algorithm, oracle/lsb_oracle.c) on this host's cores, timed on
                a bounded number of iterations of the SAME operator and scaled
                to whole solves; rank 0, N = 1 only.  CHOLMOD, the reference's
                own CPU solver, is not installable here.
and, at N = 1, one sub-record per remaining BASELINE.json config and claim:
  csr_kernel    SURVEY 8(d) to the letter: on config 3's pattern with general values,
                the kernels that stream 12 B per non-zero -- k_spmv_adaptive on the CSR
                arrays (offs / cols / vals) and k_spmv_sell with 32-bit columns --
                >= 100 back-to-back launches after >= 10 warm-ups, frac =
                (12 nnz + 20 n + 4) / t / 8 TB/s (<= 1 by construction)
  cfg2          configs[1]: tests/xn3b_A_18.txt, tol 1e-12, `trials` = 100 warm-up +
                100 timed solves (the reference's protocol, src/cholmod-impl.h:44-63),
                Jacobi-PCG and FSAI(3)-PCG, x against the golden solution, beside
                cpu_direct_baseline (cached-factor direct solve on one host core)
  cfg5          configs[4]: the power-law SpMV (two-phase form), GB/s on SURVEY's bytes
  cfg4, general_values   as before.
`--only NAME[,NAME]` runs just those sub-records (profiling); `--krylov gmres` times
GMRES(--restart) with its Gram-Schmidt passes counted in `iteration.bytes`.
"""
import argparse
import ctypes
import gzip
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402  (first: the library binds to torch's HIP/RCCL)
import torch.distributed as dist  # noqa: E402

import lsbench_amd as la  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md, chip-level parameters
WORKLOADS = {
    "lap2d": "lap2d:nx=3162,ny=3162",
    "lap2d_coef": "lap2d:nx=3162,ny=3162,coef=1",
    "lap3d": "lap3d:nx=400,ny=400,nz=400",
    "powerlaw": "powerlaw:n=8000000,gamma=1.585350372615855,max=4096,seed=20240607",
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="lap2d")
    p.add_argument("--tol", type=float, default=1e-8)
    p.add_argument("--maxit", type=int, default=100000)
    p.add_argument("--spmv", type=int, default=0, help="LSB_SPMV_* variant (0 = auto)")
    p.add_argument("--krylov", default="auto", choices=["cg", "cg1", "auto", "gmres"],
                   help="cg = classic PCG; cg1 = single-reduction PCG (2 launches, 1 reduction "
                        "per iteration); auto = cg1 when the operator spans several GPUs; gmres = "
                        "restarted GMRES(--restart), right Jacobi preconditioning")
    p.add_argument("--restart", type=int, default=30, help="GMRES restart length (1..32)")
    p.add_argument("--operator", default="upper", choices=["upper", "raw"],
                   help="file: workloads -- upper = S = triu(A) + triu(A,1)^T, the operator CHOLMOD "
                        "factorises (src/cholmod-impl.h:5-16); raw = the file matrix as it is (GMRES)")
    p.add_argument("--only", default="",
                   help="comma list of sub-records to run INSTEAD of the headline (csr_kernel, cfg2, cfg5): "
                        "one JSON line with just those -- what the PMC / trace passes profile")
    p.add_argument("--csr-kernel", type=int, default=1, help="N = 1: 1 = the line carries `csr_kernel`")
    p.add_argument("--cfg2", type=int, default=1, help="N = 1: 1 = the line carries `cfg2` (configs[1])")
    p.add_argument("--cfg2-trials", type=int, default=100)
    p.add_argument("--cfg5", type=int, default=1, help="N = 1: 1 = the line carries `cfg5` (configs[4])")
    p.add_argument("--comm", default="auto", choices=["auto", "rccl", "p2p"],
                   help="N>1: auto = direct xGMI stores where they pass the self-test and beat "
                        "RCCL; rccl / p2p force one")
    p.add_argument("--overlap", type=int, default=-1,
                   help="N>1: 1 = interior rows of the SpMV run while the halo is in flight; "
                        "-1 = only where a halo is >= 64 Ki doubles")
    p.add_argument("--spmv-tune", type=int, default=-1,
                   help="-1 = timing pass at setup picks the SpMV flavour; 0..3 force it")
    p.add_argument("--fixed-iters", type=int, default=0,
                   help="experiment mode: run exactly this many iterations per step "
                        "(tol = 0); the line is marked and is NOT a solves/s figure")
    p.add_argument("--cpu-seconds", type=float, default=12.0,
                   help="budget of the CPU baseline leg (0 = skip)")
    p.add_argument("--cfg4", type=int, default=1,
                   help="1 = every line also carries a `cfg4` sub-record: BASELINE.json configs[3] "
                        "(7-point 400^3, 64 M rows) solved --cfg4-steps times on the same N GPUs -- the "
                        "workload the 8-vs-1 GPU target is stated on; 0 = skip it")
    p.add_argument("--cfg4-steps", type=int, default=2)
    p.add_argument("--general-values", type=int, default=1,
                   help="1 = at N = 1 the line also carries `general_values`: config 3's pattern with "
                        "general values (lap2d:...,coef=1) -- the SpMV whose values must be streamed, "
                        "roofline fraction on SURVEY 8(d)'s CSR byte count; 0 = skip it")
    p.add_argument("--general-values-steps", type=int, default=1)
    p.add_argument("--persistent", type=int, default=0,
                   help="launch-bound operators: 1 = the whole solve as one persistent launch, 0 = "
                        "launch per kernel, -1 = whichever the creation-time timing finds faster")
    p.add_argument("--precond", default="jacobi", choices=["jacobi", "l1", "none", "cheb", "bj", "fsai"],
                   help="preconditioner: Jacobi (headline), l1-Jacobi, none, Chebyshev polynomial "
                        "(--cheb-degree), block-Jacobi (--block-size), factorised sparse approximate "
                        "inverse on the pattern of tril(S^k) (--fsai-power k; one GPU)")
    p.add_argument("--fsai-power", type=int, default=3)
    p.add_argument("--cheb-degree", type=int, default=4)
    p.add_argument("--block-size", type=int, default=8)
    p.add_argument("--precision", default="fp64", choices=["fp64", "fp32"],
                   help="fp32 = matrix values stored and streamed as fp32, fp64 vectors and "
                        "accumulation, fp64 iterative refinement to the same tolerance")
    p.add_argument("--verbose", type=int, default=0, help="library verbosity (2: the timing pass's lines)")
    p.add_argument("--verify", type=int, default=1,
                   help="1 = a solve counts only once ||b - S x|| <= tol ||b|| holds for the residual "
                        "recomputed from x (correction solves inside the timed region if needed)")
    return p.parse_args()


def read_file_matrix(path):
    if path.endswith(".gz"):
        # one private copy per process: the ranks of a torch.distributed.run job
        # would otherwise write and parse the same /tmp path at the same time
        fd, tmp = tempfile.mkstemp(prefix="lsb_r%s_" % os.environ.get("RANK", "0"),
                                   suffix="_" + os.path.basename(path)[:-3])
        try:
            with gzip.open(path, "rb") as fi, os.fdopen(fd, "wb") as fo:
                fo.write(fi.read())
            return la.lsbench_matrix_read(tmp)
        finally:
            os.unlink(tmp)
    return la.lsbench_matrix_read(path)


def host_cores():
    """CPU share of this process: affinity mask capped by the cgroup quota (the
    GPU box shows 256 logical CPUs but grants 16)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            parts = open(path).read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(A, gpu_iters, budget_s, is_spd):
    """Oracle PCG (OpenMP, all granted cores) on a bounded number of iterations."""
    from oracle import oracle as O  # the checker, used here only as the CPU baseline
    cores = host_cores()
    b = O.rhs(A.nrows)
    t = time.perf_counter()
    O.pcg_jacobi(A.offs, A.cols, A.vals, b, 0.0, 5, jacobi=is_spd, threads=cores)
    t5 = (time.perf_counter() - t)
    its = int(max(10, min(2000, budget_s / max(t5 / 5, 1e-6))))
    its = min(its, max(gpu_iters, 10))
    t = time.perf_counter()
    _, done, _, _ = O.pcg_jacobi(A.offs, A.cols, A.vals, b, 0.0, its, jacobi=is_spd, threads=cores)
    dt = time.perf_counter() - t
    per_iter = dt / max(done, 1)
    return {"value": 1.0 / (per_iter * max(gpu_iters, 1)), "unit": "solves/s", "cores": cores,
            "kind": "port",
            "sample": "%d OpenMP Jacobi-PCG iterations of the same operator in %.2f s "
                      "(%.3f ms/iteration), scaled to the %d iterations one GPU solve took"
                      % (done, dt, per_iter * 1e3, gpu_iters)}


def cpu_direct_baseline(A, trials=100):
    """Stand-in for the reference's CHOLMOD protocol (src/cholmod-impl.h:34-71:
    factor once untimed, `trials` warm-up + `trials` timed solves with the
    cached factor), with SciPy's SuperLU because CHOLMOD is not installable
    here.  Small file matrices only; 1 core."""
    try:
        import numpy as np
        import scipy.sparse as sp
        import scipy.sparse.linalg as sla
    except ImportError:
        return None
    n = A.nrows
    S = sp.csc_matrix(sp.csr_matrix((A.vals, A.cols.astype(np.int64), A.offs.astype(np.int64)),
                                    shape=(n, n)))
    b = np.arange(n, dtype=np.float64)
    t = time.perf_counter()
    lu = sla.splu(S, permc_spec="MMD_AT_PLUS_A", options=dict(SymmetricMode=True))
    t_factor = time.perf_counter() - t
    for _ in range(trials):
        lu.solve(b)
    t = time.perf_counter()
    for _ in range(trials):
        x = lu.solve(b)
    dt = time.perf_counter() - t
    return {"value": trials / dt, "unit": "solves/s", "cores": 1, "kind": "stand-in",
            "sample": "SuperLU (scipy %s) factor once (%.1f ms, untimed) + %d solves with the "
                      "cached factor in %.3f s; relres %.1e" % (
                          __import__("scipy").__version__, t_factor * 1e3, trials, dt,
                          float(np.linalg.norm(b - S @ x) / np.linalg.norm(b)))}


class Ctx:
    pass


def setup_ranks(a):
    c = Ctx()
    c.rank = int(os.environ.get("RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.local = int(os.environ.get("LOCAL_RANK", "0"))
    # LSB_BENCH_FORCE_DIST=1 drives the multi-rank code path (process group, RCCL
    # communicator, distributed solver constructor) with a single rank -- the
    # only way to rehearse it on a one-GPU box
    c.dist_on = c.world > 1 or os.environ.get("LSB_BENCH_FORCE_DIST") == "1"
    if c.dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
    if c.world != a.gpus:
        if c.world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = c.world
    # rehearsal switches (tests/test_dist_gpu.py): several ranks on ONE device, gloo
    # for torch.distributed; the library's own collectives then run on a test
    # double of RCCL that is LD_PRELOADed by the test
    torch.cuda.set_device(0 if os.environ.get("LSB_BENCH_ONE_GPU") == "1" else c.local)
    c.backend = os.environ.get("LSB_BENCH_BACKEND", "nccl")
    c.lib = la._lib.load()
    if la.hip_cdna4_init() != 0:
        sys.exit("hip_cdna4_init failed: no MI355X visible (there is no CPU path)")
    if c.dist_on:
        if c.backend == "nccl":
            dist.init_process_group("nccl", rank=c.rank, world_size=c.world,
                                    device_id=torch.device("cuda", c.local))
        else:
            dist.init_process_group(c.backend, rank=c.rank, world_size=c.world)
        idb = ctypes.create_string_buffer(la._lib.UNIQUE_ID_BYTES)
        if c.rank == 0:
            c.lib.lsb_hip_comm_get_unique_id(idb)
        t = torch.frombuffer(bytearray(idb.raw), dtype=torch.uint8).clone()
        t = t.cuda() if c.backend == "nccl" else t
        dist.broadcast(t, 0)
        idb = ctypes.create_string_buffer(bytes(t.cpu().tolist()), la._lib.UNIQUE_ID_BYTES)
        la._lib.check(c.lib.lsb_hip_comm_init_rank(idb, c.world, c.rank), "comm_init_rank")
    c.cdev = "cuda" if c.backend == "nccl" else "cpu"
    return c


def kernels_sha16():
    """sha256 (first 16 hex digits) of the sources the SpMV kernels and their layouts are
    built from: a committed PMC figure is only quoted for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("hip_kernels.hip", "hip_pb.hip", "lsb_operator.c"):
        with open(os.path.join(ROOT, "lsbench_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernel, value_slots, spmv_flags=None, period=None):
    """Fabric-side bytes per launch of `kernel` on `workload` from the committed PMC profile
    (profiles/pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3
    passes of this same command, tools/gpu_profiles.sh).  The bench process cannot
    read hardware counters itself, so the figure is tied to what it was measured
    on: the entry must name this kernel, the hash of the kernel / layout sources,
    the layout's kept / all value slots AND the flavour the timing pass picked
    (spmv_flags, xcd_period_slices) of THIS run -- any mismatch prints traffic null
    and says why in `traffic_source`."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ts = json.load(f).get(workload)
    except (OSError, ValueError):
        return None, "no profiles/pmc_traffic.json"
    if isinstance(ts, dict):
        ts = [ts]
    if not isinstance(ts, list) or not ts:
        return None, "no committed PMC profile of this workload"
    sha, why = kernels_sha16(), None
    for t in ts:   # one entry per profiled flavour of the workload's SpMV: the one this run's timing pass picked
        if t.get("kernel") != kernel:
            why = why or "stale: committed PMC profile is of %s, this run used %s" % (t.get("kernel"), kernel)
        elif t.get("kernels_sha16") != sha:
            why = why or ("stale: committed PMC profile was taken on kernel sources %s, these are %s"
                          % (t.get("kernels_sha16"), sha))
        elif list(t.get("value_slots", [0, 0])) != list(value_slots):
            why = why or ("stale: committed PMC profile kept %s value slots, this run %s"
                          % (t.get("value_slots"), list(value_slots)))
        elif spmv_flags is not None and (t.get("spmv_flags") != spmv_flags or t.get("xcd_period_slices") != period):
            why = ("other flavour: the committed PMC profiles are of spmv_flags / xcd_period_slices %s, this run's "
                   "timing pass picked %s / %s" % (sorted({(e.get("spmv_flags"), e.get("xcd_period_slices")) for e in ts}),
                                                   spmv_flags, period))
        else:
            return t["bytes"], t.get("source", "profiles/pmc_traffic.json")
    return None, why


FABRIC_NOTE = ("traffic = FETCH_SIZE x 2 + WRITE_SIZE of the committed rocprofv3 --pmc passes: bytes through the "
               "L2's fabric side, INCLUDING what the 256 MB Infinity Cache served (MI355X_MICROARCH.md: the "
               "counters sit in front of it) -- an upper bound of what the HBM moved, not the HBM's own count")


def gmres_bytes(n, sp, m, steps):
    """Bytes `steps` inner steps of GMRES(m) move (hip_gmres_drv.c): per step j of a cycle (basis of
    j + 1 vectors) k_gm_scale_prec 32 n (v_j and the diagonal read, v_j and z written), the SpMV's
    layout bytes, two k_gm_multidot passes of 8 n (j + 2) and two k_gm_update_w passes of 8 n (j + 3);
    per cycle k_gm_resid 24 n, k_gm_finish_cycle 8 n (steps of the cycle + 3) and, from the second
    cycle on, the copy of x into the gather vector + one more SpMV."""
    tot, done, cycle = 0, 0, 0
    while done < steps:
        k = min(m, steps - done)
        tot += 24 * n + 8 * n * (k + 3) + ((16 * n + sp) if cycle else 0)
        for j in range(k):
            tot += 32 * n + sp + 2 * 8 * n * (j + 2) + 2 * 8 * n * (j + 3)
        done += k
        cycle += 1
    return tot


def run_workload(a, c, workload, steps, warmup, cpu_leg):
    """Build the operator (this rank's rows), warm up, time EXACTLY `steps` solves
    between barriers; returns the record (rank 0's view, times = max over ranks)."""
    rank, world = c.rank, c.world
    t_setup = time.perf_counter()
    spmv_only = False
    key = workload
    if workload.startswith("file:"):
        Afile = read_file_matrix(workload[5:])
        # the operator CHOLMOD factorises, or (--operator raw, GMRES) the file matrix as it is
        A = la.lsb_csr_symmetrize_upper(Afile) if a.operator == "upper" else la.lsb_csr_copy_base0(Afile)
        n = A.nrows
        name = os.path.basename(workload[5:])
        if c.dist_on:
            b = la.lsb_csr_partition_rows(A, world)
            r0, r1 = int(b[rank]), int(b[rank + 1])
            Aloc = la.lsb_csr_row_slice(A, r0, r1)
        else:
            r0, r1, Aloc = 0, n, A
    else:
        spec = WORKLOADS.get(workload, workload)
        spmv_only = spec.startswith("powerlaw") and "spd=1" not in spec
        probe = la.lsbench_matrix_synth(spec, 0, 1)
        n = probe.n_global
        r0, r1 = (n * rank // world) & ~1, ((n * (rank + 1) // world) & ~1 if rank + 1 < world else n)
        Aloc = la.lsbench_matrix_synth(spec, r0, r1)
        name = spec
    small = Aloc.nnz < 2000000
    tol, maxit = a.tol, a.maxit
    if a.fixed_iters > 0:
        tol, maxit = 0.0, a.fixed_iters
    opts = la.default_opts(op_mode=la.OP_RAW, tol=tol, maxit=maxit, spmv_variant=a.spmv,
                           use_graph=1 if small else 0, sample_spmv=0 if small else 15,   # (odd: both kinds of k_pcg_col_px launch get sampled)
                           spmv_tune=a.spmv_tune, overlap=a.overlap,
                           comm={"auto": la.COMM_AUTO, "rccl": la.COMM_RCCL, "p2p": la.COMM_P2P}[a.comm],
                           krylov={"cg": la.KRYLOV_PCG, "cg1": la.KRYLOV_PCG1,
                                   "auto": la.KRYLOV_AUTO, "gmres": la.KRYLOV_GMRES}[a.krylov],
                           restart=a.restart,
                           precond=la.PRECOND_NONE if spmv_only else {
                               "jacobi": la.PRECOND_JACOBI, "l1": la.PRECOND_L1JACOBI,
                               "none": la.PRECOND_NONE, "cheb": la.PRECOND_CHEBYSHEV,
                               "bj": la.PRECOND_BLOCKJACOBI, "fsai": la.PRECOND_FSAI}[a.precond],
                           cheb_degree=a.cheb_degree, block_size=a.block_size, fsai_power=a.fsai_power,
                           precision=la.PREC_MIXED if a.precision == "fp32" else la.PREC_FP64,
                           persistent=a.persistent, verbose=a.verbose,
                           verify=1 if (a.verify and a.fixed_iters == 0 and a.krylov != "gmres") else 0)
    if c.dist_on:
        solver = la.Solver(Aloc, opts, row_begin=r0, n_global=n)
    else:
        solver = la.Solver(Aloc, opts)
    nl = r1 - r0
    d_b = torch.arange(r0, r1, dtype=torch.float64, device="cuda")  # b_i = i
    d_x = torch.zeros(nl, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if c.dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    nnz_loc = Aloc.nnz
    bytes_spmv = 12 * nnz_loc + 20 * nl + 4
    kernel = {la.SPMV_ADAPTIVE: "k_spmv_adaptive", la.SPMV_SUBWAVE: "k_spmv_subwave",
              la.SPMV_SCALAR: "k_spmv_scalar", la.SPMV_PANEL: "k_spmv_adaptive",
              la.SPMV_BINNED: "k_spmv_binned", la.SPMV_TWOPHASE: "k_pb_products + k_pb_reduce",
              la.SPMV_SELL: "k_spmv_sell16" if solver.spmv_flags & la.SPMV_FLAG_C16 else "k_spmv_sell"
              }.get(solver.spmv_variant, "?")
    # (the committed PMC profile is of the fp64 forms: no traffic figure for fp32 values)
    vslots = solver.sell_value_slots
    if solver.spmv_flags & la.SPMV_FLAG_TMPL and kernel == "k_spmv_sell16":
        kernel = "k_spmv_tmpl_col" if solver.spmv_flags & la.SPMV_FLAG_COL and solver.spmv_col_slices else "k_spmv_tmpl"
    traffic, traffic_src = pmc_traffic(key if world == 1 and a.precision == "fp64" else None, kernel, vslots,
                                       solver.spmv_flags, solver.spmv_period)
    # Bytes the roofline figure is quoted on.  SURVEY 8(d)'s CSR count is the figure for a
    # CSR SpMV whose values are streamed.  A layout that ELIDES values (constant slots: one
    # value per slot whose 128 entries are equal) does not move them, so for it the count
    # is what that layout must move in one launch -- its arrays as stored + x once + y once.
    layout_bytes = solver.spmv_layout_bytes
    it_bytes = solver.iteration_bytes
    single_red = solver.single_reduction
    # the two-launch iteration on a z-column plan (k_pcg_col_px): the launch that carries the SpMV also forms
    # p' = dc r + beta p and applies x += alpha p -- r and x in, p' and x out, 32 B per row on top of the layout's
    fused_p = solver.fused_p == 2
    rows_inside = solver.n_local + solver.padded   # (a line-padded grid carries its pad rows through every pass)
    spmv_kernel = kernel
    # A roofline fraction is quoted on bytes the kernel must MOVE: the layout's arrays as stored +
    # x once + y once (lsb_hip_solver_spmv_layout_bytes).  SURVEY 8(d)'s CSR count (12 B per non-zero
    # + 20 B per row) is what a CSR kernel moves; the sliced-ELL layouts move less (8 B per entry
    # where a slot is one diagonal, ONE value per constant slot), so on the CSR count their ratio to
    # peak can exceed 1 -- it is kept under csr_count, and is NOT the fraction.
    bytes_alg = layout_bytes if layout_bytes else bytes_spmv

    if spmv_only:
        # config 5: SpMV throughput only (the operator is unsymmetric)
        ms = solver.time_spmv(warmup * 10, steps * 20)
        barrier()
        gbps = bytes_spmv / ms / 1e6
        rec = {"metric": "fp64_csr_spmv_GBps", "value": gbps * world, "unit": "GB/s",
               "n_gpus": world, "steps": steps * 20, "warmup": warmup * 10,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": name, "rows": n, "nnz_per_gpu": nnz_loc},
               "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic,
                            "frac_fabric": (traffic / ms / 1e6 / HBM_PEAK_GBPS) if traffic else None,
                            "traffic_source": traffic_src, "traffic_note": FABRIC_NOTE,
                            "spmv_flags": solver.spmv_flags, "xcd_period_slices": solver.spmv_period,
                            "kernel": kernel + (" (panels)" if solver.spmv_variant == la.SPMV_PANEL else ""),
                            "launch_ms": ms, "algorithmic_bytes": bytes_spmv,
                            "bytes_basis": "SURVEY 8(d): 12 B per non-zero + 20 B per row + 4; the multi-pass forms "
                                           "(two-phase, binned) move more than that by design -- traffic says how much",
                            "value_slots": dict(zip(("kept", "all"), vslots)), "kernels_sha16": kernels_sha16(),
                            "measured": "hipEvents around %d back-to-back SpMVs" % (steps * 20)}}
        solver.destroy()
        return rec

    # ---- warm-up, then EXACTLY `steps` timed solves --------------------------
    for _ in range(warmup):
        res = solver.solve_dev(d_b, d_x)
    barrier()
    t0 = time.perf_counter()
    iters, spmv_ms, spmv_n, corr = 0, 0.0, 0, 0
    for _ in range(steps):
        res = solver.solve_dev(d_b, d_x)
        iters += res.iters
        corr += res.corrections
        spmv_ms += res.spmv_ms * res.spmv_samples
        spmv_n += res.spmv_samples
    barrier()
    dt = time.perf_counter() - t0
    dt_own = dt
    if c.dist_on:
        tt = torch.tensor([dt], dtype=torch.float64, device=c.cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    if res.status != la.STATUS_CONVERGED and not (a.fixed_iters > 0 and res.status == la.STATUS_MAXIT):
        sys.exit("solve did not converge (status %d after %d iterations): no valid number"
                 % (res.status, res.iters))
    its = iters // max(steps, 1)

    # ---- outside the timed region: ||b - S x|| / ||b|| recomputed from x, by torch ----
    d_y = torch.empty_like(d_x)
    solver.spmv_dev(d_x, d_y)
    sq = torch.stack([((d_b - d_y) ** 2).sum(), (d_b ** 2).sum()])
    if c.dist_on:
        sq = sq.to(c.cdev)
        dist.all_reduce(sq)
    true_relres = float((sq[0] / sq[1]).sqrt())
    del d_y
    if a.krylov == "gmres" and a.fixed_iters == 0 and not true_relres <= 10 * tol:
        sys.exit("GMRES: true residual %.3e against tol %.1e: no valid number" % (true_relres, tol))
    if a.verify and a.krylov != "gmres" and a.fixed_iters == 0 and not true_relres <= tol * (1 + 1e-6):
        sys.exit("true residual %.3e above tol %.1e after a verified solve: no valid number"
                 % (true_relres, tol))

    # ---- dominant kernel: SpMV, HIP events on the library's stream ---------
    if spmv_n:
        spmv_avg_ms = spmv_ms / spmv_n  # sampled inside the timed solves
        how = "hipEvent pairs around %d SpMV launches inside the timed solves" % spmv_n
        if fused_p:
            kernel = "k_pcg_col_px"
            # a run's even iterations: r p x p'' in, p' x out (48 B per row; x is updated every second iteration, with two
            # directions at once); odd ones: r p in, p' out (24): 36 on average where the layout's count has 16 (q is not stored)
            bytes_alg = layout_bytes + 20 * rows_inside
            traffic, traffic_src = pmc_traffic(key if world == 1 and a.precision == "fp64" else None, kernel, vslots,
                                               solver.spmv_flags, solver.spmv_period)
            how = ("hipEvent pairs around %d launches of k_pcg_col_px inside the timed solves (the launch that carries "
                   "the SpMV: p' = dc r + beta p, x += alpha p, q = S p', p'.q)" % spmv_n)
    else:
        fused_p = False  # (what is timed below is the plain SpMV launch)
        spmv_avg_ms = solver.time_spmv(20, 200)  # graph replay: events do not fit inside
        how = "hipEvents around 200 back-to-back launches after the timed solves"
    gbps = bytes_alg / spmv_avg_ms / 1e6
    gbps_csr = bytes_spmv / spmv_avg_ms / 1e6
    it_basis = ("SpMV layout bytes + every vector pass of the sweeps behind it, on wall-clock "
                "time per iteration (launch gaps, reductions and the stop test included)")
    if a.krylov == "gmres" and layout_bytes and iters:
        # per inner step: mean over the steps actually run (Gram-Schmidt passes grow with the basis)
        it_bytes = gmres_bytes(nl, layout_bytes, max(1, min(a.restart, 32)), iters) // iters
        it_basis = ("GMRES(%d), mean per inner step: SpMV layout bytes + k_gm_scale_prec + two k_gm_multidot "
                    "and two k_gm_update_w passes over the basis (CGS2) + the per-cycle passes, on wall-clock "
                    "time per inner step" % a.restart)
    # SURVEY.md section 8(d)'s protocol as well: >= 100 back-to-back launches after >= 10
    # warm-ups (warmer caches than inside the solve; reported, not used for `frac`)
    b2b_ms = solver.time_spmv(10, 100) if spmv_n else spmv_avg_ms
    n_tot_nnz = nnz_loc
    if c.dist_on:
        tt = torch.tensor([float(nnz_loc)], dtype=torch.float64, device=c.cdev)
        dist.all_reduce(tt)
        n_tot_nnz = int(tt.item())
    comm = dict(zip(("mode", "selftest_direct_us", "selftest_rccl_us"), solver.comm),
                modes="0 one shard, 1 RCCL, 2 direct xGMI all-reduce, 3 direct xGMI halos too")
    per_rank = None
    if c.dist_on:
        # load balance of a sharded line: every rank's own wall-clock per iteration (before the MAX),
        # its SpMV launch inside the solve, its rows and non-zeros
        mine = torch.tensor([dt_own / max(iters, 1) * 1e6, spmv_avg_ms * 1e3, float(nl), float(nnz_loc)],
                            dtype=torch.float64, device=c.cdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        per_rank = {"iteration_us": {"min": float(allr[:, 0].min()), "max": float(allr[:, 0].max())},
                    "spmv_us": {"min": float(allr[:, 1].min()), "max": float(allr[:, 1].max())},
                    "rows": [int(v) for v in allr[:, 2]], "nnz": [int(v) for v in allr[:, 3]]}
    comm.update(solver.comm_plan)  # rccl_ranks = ncclCommCount; rank 0's halo plan
    rec = {
        "metric": "cg_solves_per_sec", "value": steps / dt, "unit": "solves/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not workload.startswith("file:") else "reference tests/ matrix",
        "config": {"workload": name, "rows": n, "nnz": n_tot_nnz,
                   "solver": ("GMRES(%d), right Jacobi" % a.restart) if a.krylov == "gmres" else
                             {"jacobi": "PCG+Jacobi", "l1": "PCG+l1-Jacobi", "none": "CG", "bj": "PCG+block-Jacobi(%d)"
                              % a.block_size, "cheb": "PCG+Chebyshev(%d)" % a.cheb_degree,
                              "fsai": "PCG+FSAI(tril(S^%d))" % a.fsai_power}[a.precond] + (
                                  " (single-reduction form)" if single_red else ""),
                   "tol": tol, "rhs": "b_i=i", "partition": "row-range x%d" % world,
                   "iterations_per_solve": its, "relres": res.relres,
                   "true_relres": true_relres,
                   "stop": ("||b - S x|| <= tol ||b|| verified on the residual recomputed from x, "
                            "inside the timed region (%d correction solve(s) per solve; their "
                            "iterations are counted)" % (corr // max(steps, 1)))
                   if (a.verify and a.fixed_iters == 0 and a.krylov != "gmres") else "recurrence residual"},
        "iterations_per_sec": iters / dt,
        "setup_seconds": t_setup,
        "blas1_nt_mask": solver.blas1_nt,  # which sweep operands are loaded nontemporal (timed at creation)
        # the whole iteration against the same peak: SpMV layout bytes + 8 B per row and vector pass
        # of the sweeps (lsb_hip_solver_iteration_bytes), over the wall-clock time per iteration
        "iteration": ({"bytes": it_bytes, "us": dt / iters * 1e6, "GBps": it_bytes * iters / dt / 1e9,
                       "frac": it_bytes * iters / dt / 1e9 / HBM_PEAK_GBPS,
                       "basis": it_basis}
                      if it_bytes and iters and world == 1 else None),
        "comm": comm, "per_rank": per_rank,
        "roofline": {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": gbps / HBM_PEAK_GBPS, "traffic": traffic,
                     # fabric-side traffic (L2 <-> Infinity Cache / HBM) of the committed profile of this
                     # command, these kernel sources and this flavour / this run's launch time, over peak
                     "frac_fabric": (traffic / spmv_avg_ms / 1e6 / HBM_PEAK_GBPS) if traffic else None,
                     "traffic_source": traffic_src, "traffic_note": FABRIC_NOTE,
                     "algorithmic_bytes": bytes_alg,
                     "bytes_basis": ("layout: what the stored layout must move in one launch -- its index / code / "
                                     "slot / template arrays and the values it keeps (value_slots kept / all: one "
                                     "value per slot whose 128 entries are equal), x once, y once"
                                     + (" -- here, averaged over a run's even and odd iterations: r, p in and p' out, every second "
                                        "time also x and the direction before in and x out (q = S p' is not stored: the launch "
                                        "that updates r forms it again); the direction update and the x update ride in this "
                                        "launch (two launches and 60 instead of 88 B per row and iteration)"
                                        if fused_p else "")
                                     + "; SURVEY 8(d)'s CSR count is under csr_count (a CSR kernel's bytes: this "
                                     "layout moves fewer, so that ratio is not a fraction of anything)"
                                     if bytes_alg != bytes_spmv else
                                     "SURVEY 8(d): 12 B per non-zero + 20 B per row + 4 (values streamed)"),
                     "direction_update_in_this_launch": fused_p,
                     "csr_count": {"bytes": bytes_spmv, "GBps": gbps_csr, "ratio_to_peak": gbps_csr / HBM_PEAK_GBPS},
                     "layout_bytes": layout_bytes,
                     "value_slots": dict(zip(("kept", "all"), vslots)),
                     "kernel": kernel + (" (panels)" if solver.spmv_variant == la.SPMV_PANEL else "") + " (fused p.q)",
                     "launch_ms": spmv_avg_ms, "back_to_back_launch_ms": b2b_ms,
                     "back_to_back_kernel": ("%s (y = S x alone, %d B)" % (spmv_kernel, layout_bytes))
                     if fused_p else kernel,
                     "line_padding_rows": solver.padded,
                     "spmv_flags": solver.spmv_flags, "xcd_period_slices": solver.spmv_period,
                     "kernels_sha16": kernels_sha16(), "measured": how},
    }
    if a.fixed_iters > 0:
        rec["metric"] = "EXPERIMENT_fixed_%d_iterations_per_step" % a.fixed_iters
    if cpu_leg and rank == 0 and world == 1 and a.cpu_seconds > 0 and a.fixed_iters == 0:
        rec["cpu_baseline"] = cpu_baseline(Aloc, its, a.cpu_seconds, True)
        if workload.startswith("file:"):
            rec["cpu_direct_baseline"] = cpu_direct_baseline(Aloc)
    solver.destroy()
    del d_b, d_x
    torch.cuda.empty_cache()
    return rec


def csr_kernel_record(a, c):
    """SURVEY 8(d) to the letter, N = 1: the fp64 SpMV kernels that stream 12 B per non-zero, on
    config 3's pattern with general values (nothing for a layout to elide).  k_spmv_adaptive reads
    the CSR arrays themselves (row offsets, 32-bit columns, fp64 values -- what the reference hands
    its Krylov solver, src/ginkgo.cpp:26-34); k_spmv_sell reads the same 12 B per entry in sliced-ELL
    order (no row offsets: 16 instead of 20 B per row).  >= 100 back-to-back launches after >= 10
    warm-ups, HIP events on the library's stream; frac = (12 nnz + 20 n + 4) / t / 8 TB/s."""
    A = la.lsbench_matrix_synth(WORKLOADS["lap2d_coef"])
    n, nnz = A.nrows, A.nnz
    bytes_csr = 12 * nnz + 20 * n + 4
    out = {"workload": WORKLOADS["lap2d_coef"], "rows": n, "nnz": nnz, "algorithmic_bytes": bytes_csr,
           "bytes_basis": "SURVEY 8(d): 12 B per non-zero + 20 B per row + 4", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "warmup_launches": 10, "timed_launches": 100, "kernels_sha16": kernels_sha16(),
           "measured": "hipEvents around 100 back-to-back launches (fused p.q as in the solve) after 10 warm-ups",
           "kernels": {}}
    for name, variant, tune, grid in (("k_spmv_adaptive", la.SPMV_ADAPTIVE, -1, 0), ("k_spmv_sell", la.SPMV_SELL, 2, 1536)):
        # (adaptive: its own timing pass over {plain, prefetch, nontemporal, both}; sell: 32-bit columns
        # forced -- spmv_tune bit 2 would be the 16-bit codes)
        sv = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, spmv_variant=variant, spmv_tune=tune, spmv_grid=grid,
                                          use_graph=0, verbose=a.verbose))
        if sv.spmv_variant != variant or (variant == la.SPMV_SELL and sv.spmv_flags & la.SPMV_FLAG_C16):
            sys.exit("csr_kernel: asked for variant %d, the solver runs %d / flags %d" % (variant, sv.spmv_variant, sv.spmv_flags))
        ms = sv.time_spmv(10, 100)
        moved = sv.spmv_layout_bytes
        key = "csr_kernel:" + name
        traffic, src = pmc_traffic(key, name, sv.sell_value_slots, sv.spmv_flags, sv.spmv_period)
        out["kernels"][name] = {
            "launch_us": ms * 1e3, "achieved": bytes_csr / ms / 1e6, "frac": bytes_csr / ms / 1e6 / HBM_PEAK_GBPS,
            "layout_bytes": moved, "frac_layout": moved / ms / 1e6 / HBM_PEAK_GBPS,
            "spmv_flags": sv.spmv_flags, "grid": sv.spmv_grid, "traffic": traffic,
            "frac_fabric": (traffic / ms / 1e6 / HBM_PEAK_GBPS) if traffic else None, "traffic_source": src}
        sv.destroy()
        torch.cuda.empty_cache()
    k = out["kernels"]["k_spmv_adaptive"]
    out.update(kernel="k_spmv_adaptive (CSR arrays: offs, cols, vals)", launch_us=k["launch_us"],
               achieved=k["achieved"], frac=k["frac"], traffic=k["traffic"], traffic_note=FABRIC_NOTE)
    return out


def cfg2_record(a, c):
    """BASELINE.json configs[1], N = 1: tests/xn3b_A_18.txt (3461 rows), operator S = triu(A) + triu(A,1)^T,
    b_i = i, tol 1e-12 -- the reference's protocol (src/cholmod-impl.h:44-63): set-up untimed, `trials`
    warm-up solves, `trials` timed solves, solves/s = trials / elapsed; Jacobi-PCG (the headline
    algorithm) and FSAI(3)-PCG (set up once, like the reference's factorisation); x against the golden
    direct solution; beside it the CPU's cached-factor direct solve on the same box."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "matrices", "xn3b_A_18.txt.gz")
    A = la.lsb_csr_symmetrize_upper(read_file_matrix(path))
    n, trials = A.nrows, max(1, a.cfg2_trials)
    gold = np.fromfile(os.path.join(ROOT, "tests", "golden", "x", "xn3b_A_18.x.f64"), dtype="<f8")
    d_b = torch.arange(n, dtype=torch.float64, device="cuda")
    d_x = torch.zeros(n, dtype=torch.float64, device="cuda")
    out = {"workload": "tests/xn3b_A_18.txt", "rows": n, "nnz": A.nnz, "tol": 1e-12, "trials": trials,
           "protocol": "set-up untimed, %d warm-up + %d timed solves (src/cholmod-impl.h:44-63), x reset to 0 "
                       "before every solve" % (trials, trials), "solvers": {}}
    for name, kw in (("PCG+Jacobi", dict(precond=la.PRECOND_JACOBI)),
                     ("PCG+FSAI(tril(S^3))", dict(precond=la.PRECOND_FSAI, fsai_power=3))):
        t = time.perf_counter()
        sv = la.Solver(A, la.default_opts(op_mode=la.OP_RAW, tol=1e-12, maxit=20000, use_graph=1, persistent=0,
                                          verbose=a.verbose, **kw))
        torch.cuda.synchronize()
        t_setup = time.perf_counter() - t
        for _ in range(trials):
            res = sv.solve_dev(d_b, d_x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(trials):
            res = sv.solve_dev(d_b, d_x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if res.status != la.STATUS_CONVERGED:
            sys.exit("cfg2 %s: status %d after %d iterations" % (name, res.status, res.iters))
        err = float(np.linalg.norm(d_x.cpu().numpy() - gold) / np.linalg.norm(gold))
        if not err <= 1e-10:
            sys.exit("cfg2 %s: x is %.2e away from the golden solution (bar 1e-10)" % (name, err))
        out["solvers"][name] = {"value": trials / dt, "unit": "solves/s", "iterations_per_solve": int(res.iters),
                                "us_per_iteration": dt / trials / max(res.iters, 1) * 1e6, "relres": res.relres,
                                "err_vs_golden": err, "setup_seconds": t_setup}
        sv.destroy()
    out["value"] = out["solvers"]["PCG+Jacobi"]["value"]
    out["unit"] = "solves/s"
    out["cpu_direct_baseline"] = cpu_direct_baseline(A, trials)
    return out


def cfg5_record(a, c):
    """BASELINE.json configs[4], N = 1: the 8 M-row power-law operator (mean 32, max 4096 non-zeros
    per row, unsymmetric): SpMV only, GB/s on SURVEY 8(d)'s bytes; the form is the timing pass's
    choice (the two-phase form on every box so far)."""
    r = run_workload(a, c, "powerlaw", 3, 1, cpu_leg=False)
    out = {k: r[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step")}
    out["config"] = r["config"]
    out["spmv"] = r["roofline"]
    out["note"] = ("spmv.frac is on SURVEY 8(d)'s count (12 B per non-zero + 20 B per row); the two-phase form moves "
                   "28.2 B per non-zero by design (products written once, read once), so its ceiling on that count "
                   "is 0.41 x what its launches stream at -- DESIGN.md section 4, 'What bounds the two-phase form'")
    return out


def main():
    a = parse()
    c = setup_ranks(a)
    subs = (("csr_kernel", csr_kernel_record), ("cfg2", cfg2_record), ("cfg5", cfg5_record))
    if a.only:
        want = a.only.split(",")
        if c.world != 1 or any(w not in dict(subs) for w in want):
            sys.exit("--only takes %s, on one GPU" % ", ".join(k for k, _ in subs))
        line = {"metric": "sub-records only", "n_gpus": 1, "only": want}
        for k, f in subs:
            if k in want:
                line[k] = f(a, c)
        print(json.dumps(line), flush=True)
        return
    line = run_workload(a, c, a.workload, a.steps, a.warmup, cpu_leg=True)
    spec = WORKLOADS.get(a.workload, a.workload)
    if a.cfg4 and line["metric"] == "cg_solves_per_sec" and not spec.startswith("lap3d"):
        r4 = run_workload(a, c, "lap3d", a.cfg4_steps, 1, cpu_leg=False)
        line["cfg4"] = {k: r4[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                                           "ms_per_step", "scaling", "iterations_per_sec",
                                           "setup_seconds", "comm", "iteration")}
        line["cfg4"]["config"] = r4["config"]
        line["cfg4"]["spmv"] = {k: r4["roofline"][k] for k in ("kernel", "launch_ms", "achieved", "frac",
                                                               "algorithmic_bytes", "bytes_basis", "csr_count",
                                                               "traffic", "frac_fabric", "traffic_source",
                                                               "value_slots", "xcd_period_slices", "spmv_flags")}
        line["cfg4"]["note"] = ("BASELINE.json configs[3] on the same %d GPU(s): strong scaling of ONE 64 M-row "
                                "operator; the >= 6x target is cfg4.value(N=8) / cfg4.value(N=1)" % c.world)
    if (a.general_values and c.world == 1 and line["metric"] == "cg_solves_per_sec"
            and spec == WORKLOADS["lap2d"] and a.precision == "fp64" and a.fixed_iters == 0):
        rg = run_workload(a, c, "lap2d_coef", a.general_values_steps, 1, cpu_leg=False)
        line["general_values"] = {k: rg[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                                                     "ms_per_step", "iterations_per_sec", "setup_seconds",
                                                     "iteration")}
        line["general_values"]["config"] = rg["config"]
        line["general_values"]["spmv"] = {k: rg["roofline"][k] for k in (
            "kernel", "launch_ms", "back_to_back_launch_ms", "achieved", "peak", "unit", "frac", "algorithmic_bytes",
            "bytes_basis", "csr_count", "traffic", "frac_fabric", "traffic_source", "layout_bytes", "value_slots",
            "spmv_flags", "measured")}
        line["general_values"]["note"] = (
            "BASELINE.json configs[2]'s pattern (3162^2 5-point, 49,978,572 nnz) with GENERAL values: one hashed "
            "weight in [1/2, 3/2) per grid edge, Dirichlet, SPD (lsb_synth.c `coef=1`; stated independently in "
            "oracle/lsb_oracle.c).  Every value is streamed: this is the fp64 SpMV the metric names.  spmv.frac "
            "is on the bytes the kernel's layout moves (8 B of value per entry, no column index where a slot is "
            "one diagonal); spmv.csr_count is the same launch on SURVEY 8(d)'s CSR byte count (12 B per entry), "
            "the figure a CSR kernel would have to reach.")
    if (c.world == 1 and not c.dist_on and line["metric"] == "cg_solves_per_sec" and spec == WORKLOADS["lap2d"]
            and a.precision == "fp64" and a.fixed_iters == 0 and a.krylov != "gmres" and a.precond == "jacobi"):
        # the remaining BASELINE.json configs and the SURVEY 8(d) SpMV figure, driver-run (N = 1)
        for k, f in subs:
            if getattr(a, k):
                line[k] = f(a, c)
    if c.rank == 0:
        print(json.dumps(line), flush=True)
    if c.dist_on:
        c.lib.lsb_hip_comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
