"""Launch-bound operators as ONE persistent launch per solve (hip_persist.hip,
opts.persistent): same iterates as the single-reduction oracle, the golden
solutions of all ten reference matrices, stop rules, run-to-run identical bits,
one-XCD placement, and the creation-time choice between the two forms."""
import numpy as np
import pytest

from conftest import SPD, TOY
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", TOY + SPD)
def test_persistent_solve_reaches_golden(hip, name, matrix_path, golden_x):
    A = hip.lsbench_matrix_read(matrix_path(name))
    So = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(A.nrows)
    xo, ito, relo, sto = O.pcg1_jacobi(So.offs, So.cols, So.vals, b, 1e-12)
    xg = golden_x(name)
    s = hip.Solver(A, hip.default_opts(persistent=1))
    x, res = s.solve(b)
    x2, res2 = s.solve(b)
    s.destroy()
    if A.nrows < 4:                      # too few rows for a grid of workgroups: launches
        assert res.status == hip.STATUS_CONVERGED and np.allclose(x, xg, rtol=1e-14, atol=1e-15)
        return
    # iteration count: the classic form's, or up to ~25 % more where tol = 1e-12 sits
    # below the attainable accuracy (tj7a; summation order decides, see tests/test_pcg1.py)
    _, itc, _, _ = O.pcg_jacobi(So.offs, So.cols, So.vals, b, 1e-12)
    assert res.status == hip.STATUS_CONVERGED and itc - 3 <= int(res.iters) <= 1.25 * itc + 3
    assert res2.iters == res.iters and np.array_equal(x, x2)
    if name in TOY:
        assert np.allclose(x, xg, rtol=1e-14, atol=1e-15)
    else:
        assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
        assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-10


def test_persistent_stop_rules_and_forms(hip, matrix_path, monkeypatch):
    name = "tj7a_A_18"
    A = hip.lsbench_matrix_read(matrix_path(name))
    So = O.operator_upper(O.matrix_read(matrix_path(name)))
    b = O.rhs(A.nrows)
    # MAXIT: the oracle's iterate at exactly 17, its residual
    xo, ito, relo, sto = O.pcg1_jacobi(So.offs, So.cols, So.vals, b, 1e-12, 17)
    for env in ({}, {"LSBENCH_HIP_PERSIST_STRIDE": "8"}, {"LSBENCH_HIP_PERSIST_WGS": "8"},
                {"LSBENCH_HIP_PERSIST_WGS": "64"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = hip.Solver(A, hip.default_opts(persistent=1, maxit=17))
        x, r = s.solve(b)
        assert (r.status, r.iters, sto, ito) == (hip.STATUS_MAXIT, 17, 3, 17)
        assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-11
        assert abs(r.relres - relo) <= 1e-9 * relo
        x0, r0 = s.solve(np.zeros_like(b))               # b = 0: x = 0, nothing to do
        assert r0.status == hip.STATUS_CONVERGED and r0.iters == 0 and not x0.any()
        s.destroy()
        for k in env:
            monkeypatch.delenv(k)
    # with verification on top (correction solves run through the same launch)
    s = hip.Solver(A, hip.default_opts(persistent=1, verify=1, tol=1e-11))
    x, r = s.solve(b)
    s.destroy()
    assert r.status == 1 and 0 <= r.true_relres <= 1e-11
    true = np.linalg.norm(b - O.spmv(So.offs, So.cols, So.vals, x)) / np.linalg.norm(b)
    assert true <= 1.05e-11
    # auto: both forms timed at creation, either may win; the answer is the same
    s = hip.Solver(A, hip.default_opts(persistent=-1))
    xa, ra = s.solve(b)
    s.destroy()
    s = hip.Solver(A, hip.default_opts(persistent=0))
    xl, rl = s.solve(b)
    s.destroy()
    assert ra.status == rl.status == 1
    assert np.linalg.norm(xa - xl) / np.linalg.norm(xl) <= 1e-10
    # operators that do not qualify keep the launch form silently
    L = hip.lsbench_matrix_synth("lap2d:nx=300,ny=200")  # 60 k rows > the LDS-resident vector
    s = hip.Solver(L, hip.default_opts(op_mode=hip.OP_RAW, persistent=1, tol=1e-8))
    xl, rl = s.solve(O.rhs(L.nrows))
    s.destroy()
    assert rl.status == 1
