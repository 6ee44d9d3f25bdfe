"""opts.precision = LSB_PREC_MIXED (--precision FP32; SURVEY.md section 8(f) rank
4): matrix values streamed as fp32, vectors and every sum in fp64, fp64 iterative
refinement around it -- the SAME tolerance on the fp64 operator's residual."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SPD
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", SPD)
def test_mixed_precision_reaches_golden(hip, name, matrix_path, golden_x):
    A = hip.lsbench_matrix_read(matrix_path(name))
    S = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(A.nrows)
    xg = golden_x(name)
    s = hip.Solver(A, hip.default_opts(precision=hip.PREC_MIXED))
    x, r = s.solve(b)
    x2, r2 = s.solve(b)
    s.destroy()
    assert r.status == hip.STATUS_CONVERGED and 1 <= r.corrections <= 6
    assert 0.0 <= r.true_relres <= 1e-12 and r.relres == r.true_relres
    assert np.linalg.norm(b - O.spmv(S.offs, S.cols, S.vals, x)) / np.linalg.norm(b) <= 1.05e-12
    assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
    assert r2.iters == r.iters and np.array_equal(x, x2)


def test_mixed_precision_on_exactly_representable_values(hip):
    """the Laplacians' values are fp32 numbers: S~ = S, no refinement step is needed
    and the iterates are those of the fp64 run bit for bit; the SpMV reads less"""
    import torch
    L = hip.lsbench_matrix_synth("lap3d:nx=170,ny=160,nz=150")     # 4.08 M rows: the sliced-ELL forms
    b = O.rhs(L.nrows)
    out = {}
    for prec in (hip.PREC_FP64, hip.PREC_MIXED):
        s = hip.Solver(L, hip.default_opts(op_mode=hip.OP_RAW, precision=prec, tol=1e-9, use_graph=0))
        assert s.spmv_variant == hip.SPMV_SELL
        x, r = s.solve(b)
        ms = s.time_spmv(5, 30)
        d_y = torch.empty(L.nrows, dtype=torch.float64, device="cuda:0")
        s.spmv_dev(torch.from_numpy(b).to("cuda:0"), d_y)
        out[prec] = (x, int(r.iters), int(r.corrections), ms, d_y.cpu().numpy())
        s.destroy()
    assert out[hip.PREC_MIXED][1] == out[hip.PREC_FP64][1] and out[hip.PREC_MIXED][2] == 0
    assert np.array_equal(out[hip.PREC_MIXED][0], out[hip.PREC_FP64][0])
    assert np.array_equal(out[hip.PREC_MIXED][4], out[hip.PREC_FP64][4])
    # (fewer bytes only where values are streamed at all: a constant-coefficient grid keeps one
    # value per slot, or a 128-bit mask -- nothing left for fp32 to halve; no slower, that is all)
    assert out[hip.PREC_MIXED][3] < 1.15 * out[hip.PREC_FP64][3]


def test_mixed_precision_variants(hip, matrix_path, golden_x):
    name = "tj7a_A_15"
    A = hip.lsbench_matrix_read(matrix_path(name))
    b = O.rhs(A.nrows)
    xg = golden_x(name)
    for kw in (dict(nvirt=3), dict(krylov=hip.KRYLOV_PCG1), dict(spmv_variant=hip.SPMV_ADAPTIVE),
               dict(spmv_variant=hip.SPMV_SELL), dict(precond=hip.PRECOND_CHEBYSHEV, cheb_degree=3)):
        s = hip.Solver(A, hip.default_opts(precision=hip.PREC_MIXED, **kw))
        x, r = s.solve(b)
        s.destroy()
        assert r.status == 1 and r.true_relres <= 1e-12, kw
        assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10, kw
    # through the CLI: --precision FP32 is no longer rejected for the hip solver
    drv = os.path.join(ROOT, "lsbench_amd", "csrc", "driver")
    r = subprocess.run([drv, "--solver", "hip", "--matrix", matrix_path(name), "--trials=2",
                        "--precision=FP32", "--verbose", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    x = np.array([float(l.split("=")[1]) for l in r.stdout.splitlines() if l.startswith("x[")])
    assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
    r = subprocess.run([drv, "--solver", "hip", "--matrix", matrix_path(name), "--precision=FP16"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "not implemented" in r.stderr
