"""The direct xGMI path (lsbench_amd/csrc/hip_p2p.hip) on ONE GPU: the shards
of one process own a mailbox each and exchange halos / all-reduce dot products
by stores into each other's mailboxes -- the kernels, flags, epochs and tables
of the 8-GPU path, minus the IPC mapping (tests/test_dist_gpu.py covers that
between processes).  Oracle: the sequential Jacobi-PCG of oracle/lsb_oracle.c."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("krylov", ["PCG", "PCG1"])
@pytest.mark.parametrize("nvirt,overlap", [(2, 0), (3, 1), (8, 0), (5, 1)])
def test_direct_path_between_virtual_shards(hip, nvirt, overlap, krylov):
    L = hip.lsbench_matrix_synth("lap2d:nx=300,ny=200")
    offs, cols, vals = O.lap2d(300, 200)
    b = O.rhs(L.nrows)
    xo, ito, _, sto = O.pcg_jacobi(offs, cols, vals, b, tol=1e-10)
    s = hip.Solver(L, hip.default_opts(op_mode=hip.OP_RAW, nvirt=nvirt, overlap=overlap, tol=1e-10,
                                       krylov=getattr(hip, "KRYLOV_" + krylov), comm=hip.COMM_P2P,
                                       spmv_variant=hip.SPMV_ADAPTIVE))
    assert s.comm[0] == 3 and s.overlaps == bool(overlap)
    x, r = s.solve(b)
    x2, r2 = s.solve(b)                       # hint path, epochs keep counting
    s.destroy()
    assert r.status == 1 and abs(int(r.iters) - ito) <= 4 and r2.iters == r.iters
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8
    assert np.array_equal(x, x2)


def test_direct_path_3d_and_golden(hip, matrix_path, golden_x):
    # 3D stencil: halos of one plane (several workgroups per halo)
    L = hip.lsbench_matrix_synth("lap3d:nx=70,ny=70,nz=40")
    offs, cols, vals = O.lap3d(70, 70, 40)
    b = O.rhs(L.nrows)
    xo, ito, _, _ = O.pcg_jacobi(offs, cols, vals, b, tol=1e-10)
    s = hip.Solver(L, hip.default_opts(op_mode=hip.OP_RAW, nvirt=4, tol=1e-10, comm=hip.COMM_P2P))
    assert s.comm[0] == 3
    x, r = s.solve(b)
    s.destroy()
    assert r.status == 1 and abs(int(r.iters) - ito) <= 3
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8
    # a reference matrix (unstructured: every shard references every other)
    A = hip.lsbench_matrix_read(matrix_path("xn3b_A_12"))
    s = hip.Solver(A, hip.default_opts(nvirt=3, comm=hip.COMM_P2P))
    x, r = s.solve(O.rhs(A.nrows))
    mode = s.comm[0]
    s.destroy()
    xg = golden_x("xn3b_A_12")
    assert mode in (2, 3) and r.status == 1
    assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10


def test_maxit_and_scattered_operator(hip):
    # MAXIT in the single-reduction form: the residual fix-up all-reduces after
    # the device state has left RUNNING
    L = hip.lsbench_matrix_synth("lap2d:nx=200,ny=200")
    s = hip.Solver(L, hip.default_opts(op_mode=hip.OP_RAW, nvirt=2, maxit=23, comm=hip.COMM_P2P,
                                       krylov=hip.KRYLOV_PCG1))
    x, r = s.solve(O.rhs(L.nrows))
    s.destroy()
    offs, cols, vals = O.lap2d(200, 200)
    xo, ito, relo, sto = O.pcg1_jacobi(offs, cols, vals, O.rhs(L.nrows), tol=1e-12, maxit=23)
    assert r.status == hip.STATUS_MAXIT and r.iters == 23 and sto == 3
    assert abs(r.relres - relo) <= 1e-6 * relo


def test_overlap_is_decided_by_timing_both_forms(hip, monkeypatch):
    """opts.overlap = -1 (default): both forms of the sharded SpMV -- plain behind the exchange, or
    interior rows in front of the halo -- are TIMED on the solver's own communicator at creation
    (hip_dist.c overlap_setup) and the faster one runs; the numbers travel in the comm plan.  Round
    3's rule of thumb (split where a halo is >= 64 Ki doubles) is gone: every one-device
    measurement contradicted it.  Forced on / off and the untimed fallback still work; the
    iterates do not depend on the form."""
    big = hip.lsbench_matrix_synth("lap3d:nx=260,ny=260,nz=8")      # plane = 67600 rows
    small = hip.lsbench_matrix_synth("lap3d:nx=100,ny=100,nz=54")   # plane = 10000 rows
    for A in (big, small):
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, nvirt=2, maxit=40, tol=1e-30,
                                           spmv_variant=hip.SPMV_ADAPTIVE))
        t = s.comm_plan["overlap_timed_us"]
        assert t and t["plain"] > 0 and t["split"] > 0
        assert s.overlaps == (t["split"] < t["plain"]) == s.comm_plan["overlap"]
        x, r = s.solve(O.rhs(A.nrows))
        s.destroy()
        s1 = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, maxit=40, tol=1e-30))
        assert s1.comm_plan["overlap_timed_us"] is None and not s1.overlaps     # one shard: nothing to decide
        x1, r1 = s1.solve(O.rhs(A.nrows))
        s1.destroy()
        assert r.iters == r1.iters == 40
        assert np.linalg.norm(x - x1) / np.linalg.norm(x1) <= 1e-11
        for ov in (0, 1):                                                       # forced: no timing
            s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, nvirt=2, maxit=40, tol=1e-30, overlap=ov,
                                               spmv_variant=hip.SPMV_ADAPTIVE))
            assert s.overlaps == bool(ov) and s.comm_plan["overlap_timed_us"] is None
            x2, r2 = s.solve(O.rhs(A.nrows))
            s.destroy()
            assert r2.iters == 40 and np.linalg.norm(x2 - x1) / np.linalg.norm(x1) <= 1e-11
    monkeypatch.setenv("LSBENCH_HIP_NO_OVERLAP_TUNE", "1")
    s = hip.Solver(big, hip.default_opts(op_mode=hip.OP_RAW, nvirt=2, spmv_variant=hip.SPMV_ADAPTIVE))
    assert not s.overlaps and s.comm_plan["overlap_timed_us"] is None
    s.destroy()


@pytest.mark.parametrize("variant,nvirt,overlap", [("SELL", 4, 0), ("SELL", 3, 1), ("ADAPTIVE", 5, 1),
                                                   ("ADAPTIVE", 2, 0), ("SUBWAVE", 3, 0)])
def test_allreduce_folded_into_spmv_and_sweep(hip, monkeypatch, variant, nvirt, overlap):
    """Single-reduction CG over the direct path: the all-reduce's collect phase at
    the head of k_cg1_update (LSBENCH_HIP_AR_FOLD=1, the default), its contribute
    phase in the tail of the SpMV's last launch as well (=2) -- hip_ar.h --
    bit-identical iterates to the stand-alone all-reduce launch (=0), for the
    whole and the split (interior / boundary) SpMV; a form whose kernel cannot
    carry the tail keeps the contribute launch."""
    A = hip.lsbench_matrix_synth("lap3d:nx=48,ny=40,nz=36")
    b = O.rhs(A.nrows)
    out = {}
    for fold in ("0", "1", "2"):
        monkeypatch.setenv("LSBENCH_HIP_AR_FOLD", fold)
        s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, nvirt=nvirt, overlap=overlap, tol=1e-11,
                                           krylov=hip.KRYLOV_PCG1, comm=hip.COMM_P2P, verify=0,
                                           spmv_variant=getattr(hip, "SPMV_" + variant)))
        assert s.comm[0] == 3 and s.overlaps == bool(overlap)
        x, r = s.solve(b)
        x2, r2 = s.solve(b)            # hinted run: epochs keep counting across solves
        s.destroy()
        assert r.status == 1 and r2.iters == r.iters and np.array_equal(x, x2)
        out[fold] = (x, int(r.iters), r.relres)
    for fold in ("1", "2"):
        assert out["0"][1] == out[fold][1] and out["0"][2] == out[fold][2]
        assert np.array_equal(out["0"][0], out[fold][0])
    offs, cols, vals = O.lap3d(48, 40, 36)
    xo, ito, _, _ = O.pcg1_jacobi(offs, cols, vals, b, tol=1e-11)
    assert abs(out["1"][1] - ito) <= 3
    assert np.linalg.norm(out["1"][0] - xo) / np.linalg.norm(xo) <= 1e-9
