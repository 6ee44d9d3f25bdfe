"""Preconditioners beyond the diagonal ones:
Chebyshev polynomial in D^-1 S and block-Jacobi with dense inverted blocks.
CPU: the oracle's restatement reaches the golden solutions and cuts the outer
iterations.  GPU: the HIP path follows the oracle's iteration counts and reaches
the same solutions on all seven SPD reference matrices, alone and over shards."""
import numpy as np
import pytest

from conftest import SPD
from oracle import oracle as O


@pytest.mark.parametrize("name", SPD)
def test_oracle_preconditioners_reach_golden(name, matrix_path, golden_x):
    S = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(len(S.offs) - 1)
    xg = golden_x(name)
    _, itj, _, stj = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12)
    for kind, param in (("cheb", 2), ("cheb", 6), ("bj", 4), ("bj", 64)):
        x, it, rel, st, nsp, lmax = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, kind=kind, param=param)
        assert st == 1 and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
        if kind == "cheb":
            # a degree-m polynomial: the outer iterations (= pairs of reductions) drop by
            # about m + 1 while the multiplications by S stay within ~40 % of Jacobi's
            assert it <= 1.35 * itj / (param + 1) + 5 and nsp == it * (param + 1)
            assert nsp <= 1.5 * itj and 1.0 < lmax < 10.0
        else:
            assert it <= itj + 5 and nsp == it
    if name in ("xn3b_A_18", "tj7a_A_18"):   # (dense n^3 Cholesky in the oracle: the two smallest only)
        # one block as large as the operator = the exact inverse: one iteration, or a
        # second one where cond(S) eps is not far below the tolerance
        x, it, rel, st, nsp, _ = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, kind="bj",
                                            param=len(S.offs) - 1)
        assert st == 1 and it <= 2 and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10


@pytest.mark.parametrize("name", SPD)
def test_oracle_fsai_reaches_golden(name, matrix_path, golden_x):
    """FSAI G^T G on the pattern of tril(S^k) (oracle: breadth-first pattern, dense Gaussian
    elimination per row): golden x to 1e-10, and the iterations the set-up buys -- less than
    half of Jacobi's at k = 1, about a quarter at k = 2 (xn3b_A_18: 267 -> 115 -> 69)."""
    S = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(len(S.offs) - 1)
    xg = golden_x(name)
    _, itj, _, _ = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12)
    its = {}
    for k in (1, 2):
        x, it, rel, st, nsp, _ = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, kind="fsai", param=k)
        assert st == 1 and nsp == it and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
        its[k] = it
    assert its[1] <= 0.55 * itj and its[2] <= 0.36 * itj and its[2] < its[1]
    if name == "xn3b_A_18":
        assert (itj, its[1], its[2]) == (267, 115, 69)


@pytest.mark.parametrize("spec,power", [("lap2d:nx=23,ny=17", 1), ("lap2d:nx=23,ny=17", 2), ("lap3d:nx=7,ny=6,nz=5", 2),
                                        ("lap3d:nx=9,ny=8,nz=7", 3), ("powerlaw:n=900,avg=9,max=300,seed=3,spd=1", 2)])
def test_fsai_pattern_is_tril_of_the_power(spec, power):
    """lsb_csr_fsai_pattern (host): row i = the columns j <= i of row i of (pattern of S)^power,
    cut to the `cap` nearest the diagonal; against a scipy construction."""
    import ctypes as C
    import scipy.sparse as sp
    import lsbench_amd as la
    lib = la._lib.load()
    A = la.lsbench_matrix_synth(spec)
    n = A.nrows
    P1 = sp.csr_matrix((np.ones(A.nnz), A.cols.astype(np.int64), A.offs.astype(np.int64)), shape=(n, n))
    P1 = ((P1 + P1.T + sp.identity(n)) > 0).astype(np.float64) if "powerlaw" not in spec else P1
    Q = sp.identity(n, format="csr")
    for _ in range(power):
        Q = ((Q @ P1) > 0).astype(np.float64)
    Q = sp.tril(Q + sp.identity(n)).tocsr()
    Q.sort_indices()
    for cap in (128, 5):
        T = lib.lsb_csr_fsai_pattern(A.ptr, power, cap)
        t = T.contents
        offs = np.ctypeslib.as_array(t.offs, (n + 1,))
        cols = np.ctypeslib.as_array(t.cols, (max(int(t.nnz), 1),))
        assert t.n == n and offs[-1] == t.nnz
        for i in range(n):
            want = Q.indices[Q.indptr[i]:Q.indptr[i + 1]][-cap:]
            assert np.array_equal(cols[offs[i]:offs[i + 1]], want) and want[-1] == i
        lib.lsb_fsai_pattern_free(T)
    assert not lib.lsb_csr_fsai_pattern(A.ptr, 0, 128) and not lib.lsb_csr_fsai_pattern(A.ptr, 4, 128)


@pytest.mark.gpu
@pytest.mark.parametrize("name", SPD)
def test_hip_fsai_follows_the_oracle(hip, monkeypatch, name, matrix_path, golden_x):
    """LSB_PRECOND_FSAI: pattern on the host, rows of G by batched dense Cholesky solves on the
    device (hip_fsai.hip), z = G^T (G r) as two SpMVs.  Iteration counts of the oracle's
    restatement (Gaussian elimination on the CPU), golden x to 1e-10, repeatable, with and
    without graph replay; the iteration as three launches (the sweeps ride in the gathers of
    S, G and G^T: the default on these launch-bound operators) and as the generic six."""
    A = hip.lsbench_matrix_read(matrix_path(name))
    S = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(A.nrows)
    xg = golden_x(name)
    for power in (1, 2):
        xo, ito, relo, sto, nspo, _ = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, kind="fsai", param=power)
        its = {}
        for six in ("1", None):
            if six:
                monkeypatch.setenv("LSBENCH_HIP_NO_FSAI_FUSE", six)
            else:
                monkeypatch.delenv("LSBENCH_HIP_NO_FSAI_FUSE", raising=False)
            for graph in (0, 1):
                s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_FSAI, fsai_power=power, use_graph=graph))
                x, r = s.solve(b)
                x2, r2 = s.solve(b)
                s.destroy()
                assert r.status == hip.STATUS_CONVERGED and sto == 1
                assert abs(int(r.iters) - ito) <= max(2, ito // 25)
                assert r2.iters == r.iters and np.array_equal(x, x2)
                assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
                assert r.spmvs == r.iters                      # one multiplication by S per iteration
                its[six, graph] = int(r.iters)
        assert max(its.values()) - min(its.values()) <= 2
    # a run cut by maxit: both forms stop at the same iterate count with MAXIT
    for six in ("1", None):
        if six:
            monkeypatch.setenv("LSBENCH_HIP_NO_FSAI_FUSE", six)
        else:
            monkeypatch.delenv("LSBENCH_HIP_NO_FSAI_FUSE", raising=False)
        s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_FSAI, fsai_power=2, maxit=7, use_graph=0))
        x, r = s.solve(b)
        s.destroy()
        xo7, it7, _, st7, _, _ = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, maxit=7, kind="fsai", param=2)
        assert r.status == hip.STATUS_MAXIT and r.iters == 7 == it7
        assert np.linalg.norm(x - xo7) / np.linalg.norm(xo7) <= 1e-9


@pytest.mark.gpu
def test_hip_fsai_on_a_grid_and_its_refusals(hip, matrix_path):
    """A 3-D stencil (rows of G by wavefronts: <= 32 entries at k = 2); an indefinite operator and a
    sharded solve are refused with a message (the driver process exits non-zero)."""
    import os
    import subprocess
    from conftest import ROOT
    A = hip.lsbench_matrix_synth("lap3d:nx=24,ny=20,nz=18")
    b = O.rhs(A.nrows)
    xo, ito, _, sto, _, _ = O.pcg_prec(A.offs, A.cols, A.vals, b, 1e-10, kind="fsai", param=2)
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, precond=hip.PRECOND_FSAI, fsai_power=2, tol=1e-10))
    x, r = s.solve(b)
    s.destroy()
    assert r.status == 1 and sto == 1 and abs(int(r.iters) - ito) <= 2
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8
    drv = os.path.join(ROOT, "lsbench_amd", "csrc", "driver")
    r = subprocess.run([drv, "--solver", "hip", "--matrix", matrix_path("A0_02x02"), "--precond", "fsai", "--trials=1"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "positive definite" in r.stderr
    r = subprocess.run([drv, "--solver", "hip", "--matrix", "synth:lap2d:nx=40,ny=30", "--operator", "raw",
                        "--precond", "fsai", "--nvirt", "2", "--trials=1"], capture_output=True, text=True)
    assert r.returncode != 0 and "one shard" in r.stderr
    r = subprocess.run([drv, "--solver", "hip", "--matrix", matrix_path("xn3b_A_18"), "--precond", "fsai",
                        "--fsai-power", "2", "--trials=3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = r.stdout.splitlines()
    f = rec[rec.index("===hip_cdna4:iterations,relres,status,tol,solves_per_sec,nshards===") + 1].split(",")
    assert abs(int(f[0]) - 69) <= 3 and int(f[2]) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", SPD)
def test_hip_preconditioners_follow_the_oracle(hip, name, matrix_path, golden_x):
    A = hip.lsbench_matrix_read(matrix_path(name))
    S = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(A.nrows)
    xg = golden_x(name)
    for kind, precond, key, param in (("cheb", hip.PRECOND_CHEBYSHEV, "cheb_degree", 4),
                                      ("cheb", hip.PRECOND_CHEBYSHEV, "cheb_degree", 1),
                                      ("bj", hip.PRECOND_BLOCKJACOBI, "block_size", 8),
                                      ("bj", hip.PRECOND_BLOCKJACOBI, "block_size", 100)):
        xo, ito, relo, sto, nspo, _ = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, kind=kind, param=param)
        for graph in (0, 1):
            s = hip.Solver(A, hip.default_opts(precond=precond, use_graph=graph, **{key: param}))
            x, r = s.solve(b)
            x2, r2 = s.solve(b)
            s.destroy()
            assert r.status == hip.STATUS_CONVERGED and sto == 1
            assert abs(int(r.iters) - ito) <= max(2, ito // 25)
            assert r2.iters == r.iters and np.array_equal(x, x2)
            assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
            if kind == "cheb":
                assert r.spmvs == r.iters * (param + 1) + param


@pytest.mark.gpu
def test_hip_preconditioners_variants(hip, matrix_path, golden_x):
    name = "xn3b_A_12"
    A = hip.lsbench_matrix_read(matrix_path(name))
    S = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(A.nrows)
    xg = golden_x(name)
    # over virtual shards: the Chebyshev steps exchange halos, no reduction inside
    ref = None
    for nv, comm in ((1, hip.COMM_AUTO), (3, hip.COMM_AUTO), (4, hip.COMM_P2P)):
        s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_CHEBYSHEV, cheb_degree=3, nvirt=nv, comm=comm))
        x, r = s.solve(b)
        s.destroy()
        assert r.status == 1 and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
        ref = ref or int(r.iters)
        assert abs(int(r.iters) - ref) <= 2
    s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_BLOCKJACOBI, block_size=16, nvirt=3))
    x, r = s.solve(b)
    s.destroy()
    assert r.status == 1 and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
    # one block as large as the operator: a cached dense inverse, one or two iterations
    s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_BLOCKJACOBI, block_size=1 << 20))
    x, r = s.solve(b)
    s.destroy()
    assert r.status == 1 and r.iters <= 3 and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
    # MAXIT and b = 0 through the generic sweeps; with verification on top
    xo, ito, relo, sto, _, _ = O.pcg_prec(S.offs, S.cols, S.vals, b, 1e-12, 9, kind="cheb", param=2)
    s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_CHEBYSHEV, cheb_degree=2, maxit=9))
    x, r = s.solve(b)
    assert (r.status, r.iters, sto, ito) == (hip.STATUS_MAXIT, 9, 3, 9)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-9 and abs(r.relres - relo) <= 1e-7 * relo
    x0, r0 = s.solve(np.zeros_like(b))
    s.destroy()
    assert r0.status == 1 and r0.iters == 0 and not x0.any()
    s = hip.Solver(A, hip.default_opts(precond=hip.PRECOND_CHEBYSHEV, cheb_degree=4, verify=1, tol=1e-11))
    x, r = s.solve(b)
    s.destroy()
    assert r.status == 1 and 0 <= r.true_relres <= 1e-11
    # a stencil through the sliced-ELL SpMV and the constant diagonal
    L = hip.lsbench_matrix_synth("lap2d:nx=900,ny=700")
    offs, cols, vals = O.lap2d(900, 700)
    bl = O.rhs(L.nrows)
    xo, ito, _, sto, nspo, _ = O.pcg_prec(offs, cols, vals, bl, 1e-9, kind="cheb", param=4)
    s = hip.Solver(L, hip.default_opts(op_mode=hip.OP_RAW, precond=hip.PRECOND_CHEBYSHEV, cheb_degree=4,
                                       tol=1e-9, use_graph=0))
    x, r = s.solve(bl)
    s.destroy()
    assert r.status == 1 and abs(int(r.iters) - ito) <= 3
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-7


@pytest.mark.gpu
def test_preconditioners_through_the_driver(hip, matrix_path, golden_x):
    """`driver --solver hip --precond bj --block-size N` / `--precond cheb --cheb-degree M`:
    the options behind the reference's CLI; one block as large as the operator is the
    cached dense inverse (one iteration)."""
    import os
    import subprocess
    from conftest import ROOT
    drv = os.path.join(ROOT, "lsbench_amd", "csrc", "driver")
    xg = golden_x("xn3b_A_18")
    for extra, itmax in ((["--precond", "bj", "--block-size", "100000"], 2),
                         (["--precond", "cheb", "--cheb-degree", "4"], 70)):
        r = subprocess.run([drv, "--solver", "hip", "--matrix", matrix_path("xn3b_A_18"), "--trials=3",
                            "--verbose", "2"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = r.stdout.splitlines()
        h = lines[lines.index("===hip_cdna4:iterations,relres,status,tol,solves_per_sec,nshards===") + 1].split(",")
        x = np.array([float(l.split("=")[1]) for l in lines if l.startswith("x[")])
        assert int(h[2]) == 1 and int(h[0]) <= itmax
        assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("degree,precision", [(1, "FP64"), (3, "FP64"), (4, "FP64"), (4, "MIXED")])
def test_chebyshev_steps_in_the_spmv_epilogue(hip, monkeypatch, degree, precision):
    """One shard in the 16-bit sliced-ELL form: the Chebyshev steps ride in the SpMV's
    epilogue (k_spmv_sell16<.., CHEB>; S z is never written, z' ping-pongs between two
    gather vectors) -- the same expression as k_cheb_step, so the solve is bit-identical
    to the one with the steps as launches of their own (LSBENCH_HIP_CHEB_FUSE=0), odd and
    even degree, launches and graph replay, fp64 and fp32 matrix values (the stencil's
    values are exact in fp32: the same iterates); and it follows the oracle like that one."""
    A = hip.lsbench_matrix_synth("lap2d:nx=301,ny=187")      # odd row count: the last lane's single row
    b = O.rhs(A.nrows)
    offs, cols, vals = O.lap2d(301, 187)
    xo, ito, _, sto, _, _ = O.pcg_prec(offs, cols, vals, b, 1e-11, kind="cheb", param=degree)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("LSBENCH_HIP_CHEB_FUSE", fuse)
        for graph in (0, 1):
            s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, precond=hip.PRECOND_CHEBYSHEV,
                                               cheb_degree=degree, tol=1e-11, use_graph=graph,
                                               spmv_variant=hip.SPMV_SELL,
                                               precision=getattr(hip, "PREC_" + precision)))
            assert s.spmv_variant == hip.SPMV_SELL
            x, r = s.solve(b)
            x2, r2 = s.solve(b)
            s.destroy()
            assert r.status == 1 and r2.iters == r.iters and np.array_equal(x, x2)
            if precision == "FP64":   # (the refinement rounds of mixed precision recompute residuals)
                assert r.spmvs == r.iters * (degree + 1) + degree
            out[fuse, graph] = (x, int(r.iters), r.relres)
    ref = out["0", 0]
    for k, v in out.items():
        assert v[1] == ref[1] and v[2] == ref[2] and np.array_equal(v[0], ref[0]), k
    assert sto == 1 and abs(ref[1] - ito) <= max(2, ito // 25)
    assert np.linalg.norm(ref[0] - xo) / np.linalg.norm(xo) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("nvirt,comm", [(3, "COMM_AUTO"), (4, "COMM_P2P")])
def test_chebyshev_epilogue_over_shards(hip, monkeypatch, nvirt, comm):
    """The same over shards: the halo of z is exchanged in front of every step (from the
    gather vector the step reads), the epilogue writes z' of the shard's own rows into the
    other one -- bit-identical to the steps as launches, whatever carries the exchange."""
    A = hip.lsbench_matrix_synth("lap3d:nx=40,ny=36,nz=30")
    b = O.rhs(A.nrows)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("LSBENCH_HIP_CHEB_FUSE", fuse)
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, precond=hip.PRECOND_CHEBYSHEV, cheb_degree=3,
                                           tol=1e-11, nvirt=nvirt, comm=getattr(hip, comm), verify=0,
                                           spmv_variant=hip.SPMV_SELL))
        x, r = s.solve(b)
        x2, r2 = s.solve(b)
        mode = s.comm[0]
        s.destroy()
        assert r.status == 1 and r2.iters == r.iters and np.array_equal(x, x2)
        out[fuse] = (x, int(r.iters), r.relres, mode)
    if out["0"][3] == out["1"][3]:
        assert out["0"][1] == out["1"][1] and out["0"][2] == out["1"][2]
        assert np.array_equal(out["0"][0], out["1"][0])
    else:
        # COMM_AUTO times both transports at creation and the two solvers took different ones: the all-reduce
        # adds the shards' sums in another order -- the same solve to rounding, not to the bit
        assert abs(out["0"][1] - out["1"][1]) <= 1 and abs(out["0"][2] - out["1"][2]) <= 1e-9 * out["0"][2]
        assert np.linalg.norm(out["0"][0] - out["1"][0]) <= 1e-12 * np.linalg.norm(out["0"][0])
    offs, cols, vals = O.lap3d(40, 36, 30)
    xo, ito, _, sto, _, _ = O.pcg_prec(offs, cols, vals, b, 1e-11, kind="cheb", param=3)
    assert sto == 1 and abs(out["1"][1] - ito) <= max(2, ito // 25)
    assert np.linalg.norm(out["1"][0] - xo) / np.linalg.norm(xo) <= 1e-9
