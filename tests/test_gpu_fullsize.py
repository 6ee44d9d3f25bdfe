"""BASELINE.json configs 3-5 at FULL size on the MI355X: the oracle's CPU SpMV
is fast enough to check every element, and size-independent properties cover
the solves (true residual with an independent SpMV, A*1 = boundary indicator,
linearity, identical early iterates)."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
GAMMA = 1.585350372615855


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _spmv_check(hip, A, s, x, threads):
    import torch
    d_y = torch.full((A.nrows,), float("nan"), dtype=torch.float64, device="cuda:0")
    s.spmv_dev(_dev(x), d_y)
    y = d_y.cpu().numpy()
    yo = O.spmv(A.offs, A.cols, A.vals, x, threads=threads)
    bound = 4 * EPS * np.maximum(np.diff(A.offs.astype(np.int64)), 1) * \
        O.spmv(A.offs, A.cols, np.abs(A.vals), np.abs(x), threads=threads)
    assert np.all(np.abs(y - yo) <= bound)
    return y


def test_lap2d_10m_rows(hip):
    """config 3: 3162 x 3162 5-point Laplacian, n = 9,998,244, nnz = 49,978,572."""
    import torch
    thr = min(O.max_threads(), 16)
    A = hip.lsbench_matrix_synth("lap2d:nx=3162,ny=3162")
    n = A.nrows
    assert (n, A.nnz) == (9998244, 49978572)
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=1e-8, use_graph=0))
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    y = _spmv_check(hip, A, s, x, thr)
    # A*1: 4 - (number of neighbours) exactly
    d_y = torch.empty(n, dtype=torch.float64, device="cuda:0")
    s.spmv_dev(torch.ones(n, dtype=torch.float64, device="cuda:0"), d_y)
    y1 = d_y.cpu().numpy().reshape(3162, 3162)
    assert y1[1:-1, 1:-1].max() == 0 and y1[0, 0] == 2 and y1[0, 5] == 1 and y1[-1, -1] == 2
    # linearity: A(2x - 3z) = 2Ax - 3Az to round-off
    z = rng.standard_normal(n)
    s.spmv_dev(_dev(z), d_y)
    yz = d_y.cpu().numpy()
    s.spmv_dev(_dev(2 * x - 3 * z), d_y)
    assert np.allclose(d_y.cpu().numpy(), 2 * y - 3 * yz, rtol=0, atol=1e-12 * 50)
    # early iterates equal the oracle's (same recurrences, different summation order)
    b = O.rhs(n)
    s60 = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=0.0, maxit=60, use_graph=0))
    x60, r60 = s60.solve(b)
    s60.destroy()
    xo, ito, relo, sto = O.pcg_jacobi(A.offs, A.cols, A.vals, b, 0.0, 60, threads=thr)
    assert r60.iters == 60 and r60.status == hip.STATUS_MAXIT and ito == 60
    assert np.linalg.norm(x60 - xo) / np.linalg.norm(xo) <= 1e-10
    assert abs(r60.relres - relo) <= 1e-9 * relo
    # the bench's solve: tol 1e-8, true residual from the oracle's SpMV
    d_b, d_x = _dev(b), torch.empty(n, dtype=torch.float64, device="cuda:0")
    res = s.solve_dev(d_b, d_x)
    xs = d_x.cpu().numpy()
    assert res.status == hip.STATUS_CONVERGED and res.relres <= 1e-8
    true = np.linalg.norm(b - O.spmv(A.offs, A.cols, A.vals, xs, threads=thr)) / np.linalg.norm(b)
    assert true <= 2e-8
    res2 = s.solve_dev(d_b, d_x)
    assert res2.iters == res.iters and np.array_equal(d_x.cpu().numpy(), xs)  # deterministic
    s.destroy()
    # the sharded forms at full size: 8 row-range shards on the one device, the
    # multi-GPU defaults (single-reduction PCG), both transports; 60 iterations
    xo1, it1, rel1, _ = O.pcg1_jacobi(A.offs, A.cols, A.vals, b, 0.0, 60)
    for comm in (hip.COMM_RCCL, hip.COMM_P2P):
        sp = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=0.0, maxit=60, nvirt=8, comm=comm,
                                            krylov=hip.KRYLOV_AUTO))
        assert sp.comm[0] == (1 if comm == hip.COMM_RCCL else 3)
        xp, rp = sp.solve(b)
        sp.destroy()
        assert rp.iters == 60 and rp.status == hip.STATUS_MAXIT
        assert np.linalg.norm(xp - xo1) / np.linalg.norm(xo1) <= 1e-10
        assert abs(rp.relres - rel1) <= 1e-9 * rel1


def test_lap2d_10m_rows_general_values(hip):
    """config 3's pattern with GENERAL values (`coef=1`: one hashed weight per grid edge) --
    the operator the roofline line "fp64 CSR SpMV whose values must be streamed" is measured
    on.  The product generator against the oracle's independent statement at full size, bit
    for bit; every element of the SpMV; nothing elided by the layout; the oracle's early
    iterates; the bench's solve with the true residual from the oracle's SpMV."""
    import torch
    thr = min(O.max_threads(), 16)
    A = hip.lsbench_matrix_synth("lap2d:nx=3162,ny=3162,coef=1")
    n = A.nrows
    assert (n, A.nnz) == (9998244, 49978572)
    oo, oc, ov = O.lap_coef(3162, 3162, None, 1)
    assert np.array_equal(A.offs, oo) and np.array_equal(A.cols, oc) and np.array_equal(A.vals, ov)
    del oo, oc, ov
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=1e-8, use_graph=0))
    kept, total = s.sell_value_slots
    assert s.spmv_variant == hip.SPMV_SELL and kept == total > 0      # every value is streamed
    assert s.spmv_layout_bytes >= 8 * A.nnz + 16 * n
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    _spmv_check(hip, A, s, x, thr)
    b = O.rhs(n)
    s60 = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=0.0, maxit=60, use_graph=0))
    x60, r60 = s60.solve(b)
    s60.destroy()
    xo, ito, relo, sto = O.pcg_jacobi(A.offs, A.cols, A.vals, b, 0.0, 60, threads=thr)
    assert r60.iters == 60 and r60.status == hip.STATUS_MAXIT and ito == 60
    assert np.linalg.norm(x60 - xo) / np.linalg.norm(xo) <= 1e-10
    assert abs(r60.relres - relo) <= 1e-9 * relo
    d_b, d_x = _dev(b), torch.empty(n, dtype=torch.float64, device="cuda:0")
    res = s.solve_dev(d_b, d_x)
    xs = d_x.cpu().numpy()
    assert res.status == hip.STATUS_CONVERGED and res.relres <= 1e-8
    true = np.linalg.norm(b - O.spmv(A.offs, A.cols, A.vals, xs, threads=thr)) / np.linalg.norm(b)
    assert true <= 2e-8
    res2 = s.solve_dev(d_b, d_x)
    assert res2.iters == res.iters and np.array_equal(d_x.cpu().numpy(), xs)  # deterministic
    s.destroy()


def test_powerlaw_8m_rows_spmv(hip):
    """config 5: 8M rows, mean 32 / max 4096 nnz per row (load-balance stress)."""
    thr = min(O.max_threads(), 16)
    A = hip.lsbench_matrix_synth("powerlaw:n=8000000,gamma=%r,max=4096,seed=20240607" % GAMMA)
    d = np.diff(A.offs.astype(np.int64))
    assert d.max() <= 4096 and d.min() >= 1 and 31 < d.mean() < 33
    x = np.random.default_rng(1).standard_normal(A.nrows)
    for variant in (hip.SPMV_TWOPHASE, hip.SPMV_BINNED, hip.SPMV_ADAPTIVE, hip.SPMV_SUBWAVE):
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, precond=hip.PRECOND_NONE,
                                           spmv_variant=variant))
        _spmv_check(hip, A, s, x, thr)
        s.destroy()


def test_lap3d_64m_rows(hip):
    """config 4 on ONE device: 400^3 7-point Laplacian, n = 64e6, nnz = 447,040,000."""
    import torch
    thr = min(O.max_threads(), 16)
    A = hip.lsbench_matrix_synth("lap3d:nx=400,ny=400,nz=400")
    n = A.nrows
    assert (n, A.nnz) == (64000000, 447040000)
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=1e-6, use_graph=0))
    x = np.random.default_rng(2).standard_normal(n)
    _spmv_check(hip, A, s, x, thr)
    b = O.rhs(n)
    d_b, d_x = _dev(b), torch.empty(n, dtype=torch.float64, device="cuda:0")
    res = s.solve_dev(d_b, d_x)
    assert res.status == hip.STATUS_CONVERGED
    xs = d_x.cpu().numpy()
    true = np.linalg.norm(b - O.spmv(A.offs, A.cols, A.vals, xs, threads=thr)) / np.linalg.norm(b)
    assert true <= 2e-6
    s.destroy()
    del d_b, d_x
    torch.cuda.empty_cache()
    # the z-column walk (k_spmv_tmpl_col) at full size: planes 1 .. 398 in columns but for the 8 slices of
    # every plane that hold rows of its first / last grid line, y bit for bit what the template kernel stores
    d_xr = _dev(x)
    ys = []
    for tune in (6 | 64, 6 | 64 | 256):
        st = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, spmv_variant=hip.SPMV_SELL, spmv_tune=tune, use_graph=0))
        assert st.spmv_flags == tune and st.spmv_col_slices == 398 * (1250 - 8)
        d_y = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda:0")
        st.spmv_dev(d_xr, d_y)
        ys.append(d_y)
        st.destroy()
    assert torch.equal(ys[0], ys[1])
    del ys, d_xr, d_y
    torch.cuda.empty_cache()
    # config 4 AS BASELINE.json states it: row-partitioned 8 ways (50 planes of
    # 400 x 400 per shard, one 1.28 MB plane per neighbour and exchange).  Eight
    # row-range shards on the one device run the multi-GPU defaults (single-
    # reduction PCG) over both transports; 30 iterations against the oracle's.
    xo1, it1, rel1, _ = O.pcg1_jacobi(A.offs, A.cols, A.vals, b, 0.0, 30)
    for comm in (hip.COMM_RCCL, hip.COMM_P2P):
        sp = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=0.0, maxit=30, nvirt=8, comm=comm,
                                            krylov=hip.KRYLOV_AUTO))
        assert sp.comm[0] == (1 if comm == hip.COMM_RCCL else 3)
        xp, rp = sp.solve(b)
        sp.destroy()
        assert rp.iters == 30 and rp.status == hip.STATUS_MAXIT and it1 == 30
        assert np.linalg.norm(xp - xo1) / np.linalg.norm(xo1) <= 1e-10
        assert abs(rp.relres - rel1) <= 1e-9 * rel1


def test_bench_line_contract():
    """`python bench.py` end to end at full size: ONE JSON line with the contract's keys, the
    roofline and cpu_baseline objects, a solve that is converged on the residual recomputed from
    x -- and a sub-record for every other BASELINE.json config (cfg2, cfg4, cfg5; general_values)
    plus csr_kernel, the SpMV figure that follows SURVEY 8(d) to the letter."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
                        "--cpu-seconds", "3", "--cfg4-steps", "1"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "cfg4",
              "general_values", "csr_kernel", "cfg2", "cfg5"):
        assert k in d, k
    assert d["metric"] == "cg_solves_per_sec" and d["unit"] == "solves/s" and d["n_gpus"] == 1
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["config"]["workload"].startswith("lap2d:nx=3162") and d["config"]["rows"] == 9998244
    assert d["config"]["true_relres"] <= d["config"]["tol"] * (1 + 1e-6)
    assert abs(d["value"] - d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) <= 1e-9 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    # A roofline fraction is a fraction: on this constant-coefficient operator the sliced-ELL
    # form keeps ONE value per slot of equal values, so the figure is quoted on the bytes that
    # layout must move (x once, y once, slot records, kept values) -- never on SURVEY's CSR
    # count, which it does not move (that ratio stays in the record as csr_count, > 1 allowed).
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.2 < rf["frac"] <= 1.0
    assert 0 < rf["value_slots"]["kept"] < rf["value_slots"]["all"] // 8
    csr = 12 * 49978572 + 20 * 9998244 + 4
    # the grid's lines (3162 rows) are padded to 25 whole slices inside the solver; the iteration is the
    # two-launch form on the z-column plan along y, and the launch that carries the SpMV (k_pcg_col_px) also
    # reads r (and every second time x and the direction before) and writes p' (and x) instead of q: 20 B per row on
    # average (pad rows included) on top of the layout's bytes
    rows_in = 9998244 + rf["line_padding_rows"]
    assert rf["line_padding_rows"] == 3162 * 38 and rf["kernel"].startswith("k_pcg_col_px")
    assert rf["direction_update_in_this_launch"] is True and rf["back_to_back_kernel"].startswith("k_spmv_tmpl_col")
    assert rf["csr_count"]["bytes"] == csr and rf["layout_bytes"] < csr // 3
    assert rf["algorithmic_bytes"] == rf["layout_bytes"] + 20 * rows_in     # 36 B per row on average; q is not stored
    assert rf["layout_bytes"] >= 16 * rows_in                 # x once + y once at the very least
    assert rf["back_to_back_launch_ms"] < rf["launch_ms"]     # (the SpMV alone)
    assert rf["traffic"] is None or (0.3 < rf["frac_fabric"] <= 1.0 and rf["traffic"] >= 0.9 * rf["algorithmic_bytes"])
    assert "frac_hbm" not in rf and "Infinity Cache" in rf["traffic_note"]
    assert rf["traffic_source"] and len(rf["kernels_sha16"]) == 16
    assert d["comm"]["rccl_ranks"] == 0 and d["comm"]["recv_peers"] == 0   # one shard: no communicator
    # the whole iteration on the same peak: the layout's matrix-side bytes (twice) + 7.5 vector passes (k_pcg_col_px:
    # r p in, p' out, every second time also x p'' in, x out; k_pcg_col_r: p' r in, r out), over wall-clock time per
    # iteration -- a fraction too
    it = d["iteration"]
    assert it["bytes"] == 2 * (rf["layout_bytes"] - 16 * rows_in) + 60 * rows_in and 0.3 < it["frac"] <= 1.0
    assert abs(it["frac"] - it["bytes"] / it["us"] / 1e3 / 8000.0) < 1e-9
    # the thing the metric names: the same pattern with general values, every value streamed.  The
    # fraction is on the bytes the layout moves (8 B of value per entry, no column index on a
    # diagonal slot) -- <= 1 by construction, target >= 0.6; the same launch on SURVEY 8(d)'s CSR
    # byte count sits under csr_count (it came out at 1.01 of peak: a ratio, not a fraction)
    g = d["general_values"]
    assert g["config"]["workload"].endswith("coef=1") and g["config"]["nnz"] == 49978572
    assert g["config"]["true_relres"] <= g["config"]["tol"] * (1 + 1e-6) and g["value"] > 0
    gs = g["spmv"]
    assert gs["csr_count"]["bytes"] == csr and gs["value_slots"]["kept"] == gs["value_slots"]["all"] > 0
    assert 8 * 49978572 + 16 * 9998244 <= gs["algorithmic_bytes"] == gs["layout_bytes"] < csr
    assert abs(gs["frac"] - gs["achieved"] / 8000.0) < 1e-12 and 0.6 <= gs["frac"] <= 1.0
    assert gs["csr_count"]["ratio_to_peak"] >= 0.85           # a CSR kernel would need this rate
    assert g["iteration"]["bytes"] == gs["layout_bytes"] + 88 * 9998244     # + the diagonal, read twice
    assert 0.3 < g["iteration"]["frac"] <= 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    c4 = d["cfg4"]
    assert c4["config"]["rows"] == 64000000 and c4["config"]["nnz"] == 447040000 and c4["value"] > 0
    assert c4["config"]["true_relres"] <= c4["config"]["tol"] * (1 + 1e-6) and c4["n_gpus"] == 1
    assert "pcg_iteration_GBps" not in d
    # SURVEY 8(d) to the letter: kernels that stream 12 B per non-zero, on the CSR byte count, after
    # >= 10 warm-ups over >= 100 launches -- a fraction (<= 1) by construction, target >= 0.6
    ck = d["csr_kernel"]
    assert ck["algorithmic_bytes"] == csr and ck["workload"].endswith("coef=1") and ck["timed_launches"] >= 100
    assert ck["warmup_launches"] >= 10 and ck["kernel"].startswith("k_spmv_adaptive")
    assert abs(ck["frac"] - csr / ck["launch_us"] / 1e3 / 8000.0) < 1e-9 and 0.55 <= ck["frac"] <= 1.0
    for name, k in ck["kernels"].items():
        assert 0.55 <= k["frac"] <= 1.0 and k["layout_bytes"] >= 12 * 49978572 + 16 * 9998244, name
        assert k["layout_bytes"] <= csr + 8 * 9998244 and not k["spmv_flags"] & 4          # 32-bit columns
    assert set(ck["kernels"]) == {"k_spmv_adaptive", "k_spmv_sell"}
    # configs[1]: the reference's protocol on its own test matrix, x against the golden direct solve
    c2 = d["cfg2"]
    assert c2["rows"] == 3461 and c2["tol"] == 1e-12 and c2["trials"] == 100
    for name in ("PCG+Jacobi", "PCG+FSAI(tril(S^3))"):
        sv = c2["solvers"][name]
        assert sv["value"] > 50 and sv["err_vs_golden"] <= 1e-10 and sv["relres"] <= 1e-12
    assert abs(c2["solvers"]["PCG+Jacobi"]["iterations_per_solve"] - 267) <= 2
    assert c2["solvers"]["PCG+FSAI(tril(S^3))"]["iterations_per_solve"] < 80
    assert c2["cpu_direct_baseline"]["cores"] == 1 and c2["cpu_direct_baseline"]["value"] > 0
    # configs[4]: the power-law SpMV on SURVEY's bytes
    c5 = d["cfg5"]
    assert c5["metric"] == "fp64_csr_spmv_GBps" and c5["config"]["rows"] == 8000000
    assert c5["spmv"]["algorithmic_bytes"] == 12 * c5["config"]["nnz_per_gpu"] + 20 * 8000000 + 4
    assert 0.1 < c5["spmv"]["frac"] <= 1.0 and abs(c5["value"] - c5["spmv"]["achieved"]) < 1e-6 * c5["value"]


def test_powerlaw_spd_8m_rows_cg(hip):
    """config 5 as SURVEY 8(d) defines it for CG runs: S = (B + B^T) + diag(1 + sum|row|)
    on the 8 M-row power-law structure (521 M non-zeros), Jacobi-PCG through the SpMV
    the timing pass picks for scattered rows; the oracle's iterates and stop."""
    import torch
    thr = min(O.max_threads(), 16)
    A = hip.lsbench_matrix_synth("powerlaw:n=8000000,gamma=%r,max=4096,seed=20240607,spd=1" % GAMMA)
    assert A.nrows == 8000000 and 5.0e8 < A.nnz < 5.4e8
    b = O.rhs(A.nrows)
    xo, ito, relo, sto = O.pcg_jacobi(A.offs, A.cols, A.vals, b, 1e-10, threads=thr)
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, tol=1e-10, use_graph=0, verify=1))
    assert s.spmv_variant in (hip.SPMV_BINNED, hip.SPMV_TWOPHASE, hip.SPMV_ADAPTIVE)
    d_b, d_x = _dev(b), torch.empty(A.nrows, dtype=torch.float64, device="cuda:0")
    res = s.solve_dev(d_b, d_x)
    x = d_x.cpu().numpy()
    s.destroy()
    assert sto == 1 and res.status == hip.STATUS_CONVERGED and abs(int(res.iters) - ito) <= 1
    assert 0 <= res.true_relres <= 1e-10
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-9
