"""GMRES(m): the oracle's restatement against scipy
on the CPU, the HIP path against the oracle and a sparse direct solve on the
GPU.  Operators: a convection-diffusion stencil (unsymmetric), the RAW file
matrix of the reference (symmetric only to 1e-7), a diagonally dominant
power-law matrix."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as sla

from oracle import oracle as O

GAMMA = 1.585350372615855


def convdiff(nx, ny, c=0.6):
    """5-point diffusion + first-order upwind convection: unsymmetric, M-matrix."""
    def t(n, lo, hi):
        return sp.diags([lo, 2.0, hi], [-1, 0, 1], shape=(n, n))
    A = sp.kron(sp.eye(ny), t(nx, -1 - c, -1 + c)) + sp.kron(t(ny, -1 - c / 2, -1 + c / 2), sp.eye(nx))
    A = A.tocsr()
    A.sort_indices()
    return A


def dominant_powerlaw(n, seed):
    thr, _ = O.powerlaw_table(GAMMA, 256)
    o, c, v = O.powerlaw(n, thr, seed)
    B = sp.csr_matrix((v, c, o.astype(np.int64)), shape=(n, n))
    A = (B + sp.diags(1.0 + np.asarray(abs(B).sum(axis=1)).ravel())).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def operators():
    return {"convdiff": convdiff(60, 45), "powerlaw": dominant_powerlaw(4000, 3)}


@pytest.mark.parametrize("name", ["convdiff", "powerlaw"])
@pytest.mark.parametrize("restart", [5, 30])
def test_oracle_gmres_vs_direct(name, restart):
    A = operators()[name]
    n = A.shape[0]
    b = O.rhs(n)
    x, it, rel, st = O.gmres_jacobi(A.indptr, A.indices, A.data, b, 1e-10, 5000, restart)
    xd = sla.spsolve(A.tocsc(), b)
    assert st == 1 and rel <= 1e-10
    assert np.linalg.norm(b - A @ x) / np.linalg.norm(b) <= 2e-10  # estimate == true residual
    assert np.linalg.norm(x - xd) / np.linalg.norm(xd) <= 1e-7
    # a symmetric positive definite operator: GMRES agrees with the PCG oracle
    o, c, v = O.lap2d(20, 17)
    xg, itg, _, stg = O.gmres_jacobi(o, c, v, O.rhs(340), 1e-11, 5000, 30)
    xc, _, _, _ = O.pcg_jacobi(o, c, v, O.rhs(340), 1e-12)
    assert stg == 1 and np.linalg.norm(xg - xc) / np.linalg.norm(xc) <= 1e-9


def test_oracle_gmres_stop_rules():
    A = operators()["convdiff"]
    b = O.rhs(A.shape[0])
    x, it, rel, st = O.gmres_jacobi(A.indptr, A.indices, A.data, b, 1e-14, 17, 5)
    assert st == 3 and it == 17
    x, it, rel, st = O.gmres_jacobi(A.indptr, A.indices, A.data, 0 * b, 1e-10, 100, 5)
    assert st == 1 and it == 0 and not x.any()


# ---------------------------------------------------------------------------- GPU

def _matrix(hip, A):
    return hip.Matrix.from_arrays(A.indptr, A.indices, A.data)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["convdiff", "powerlaw"])
@pytest.mark.parametrize("restart", [5, 30, 32])
def test_hip_gmres_matches_oracle(hip, name, restart):
    A = operators()[name]
    n = A.shape[0]
    b = O.rhs(n)
    xo, ito, relo, sto = O.gmres_jacobi(A.indptr, A.indices, A.data, b, 1e-10, 5000, restart)
    s = hip.Solver(_matrix(hip, A), hip.default_opts(op_mode=hip.OP_RAW, krylov=hip.KRYLOV_GMRES,
                                                     restart=restart, tol=1e-10, maxit=5000))
    x, res = s.solve(b)
    x2, res2 = s.solve(b)  # state is reset between solves
    s.destroy()
    assert res.status == hip.STATUS_CONVERGED and res.relres <= 1e-10
    assert abs(int(res.iters) - ito) <= max(2, ito // 50)
    assert res2.iters == res.iters and np.array_equal(x, x2)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-7
    assert np.linalg.norm(b - A @ x) / np.linalg.norm(b) <= 2e-10
    xd = sla.spsolve(A.tocsc(), b)
    assert np.linalg.norm(x - xd) / np.linalg.norm(xd) <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("name,nvirt", [("convdiff", 2), ("convdiff", 5), ("powerlaw", 3)])
def test_hip_gmres_over_row_range_shards(hip, name, nvirt):
    """GMRES(m) with the operator split into row-range shards: halo exchange in
    front of every SpMV, ONE all-reduce per Gram-Schmidt pass (all j+1
    coefficients together) and one per norm; every shard keeps its own copy of
    the small state and must take the same decisions."""
    A = operators()[name]
    b = O.rhs(A.shape[0])
    M = _matrix(hip, A)
    kw = dict(op_mode=hip.OP_RAW, krylov=hip.KRYLOV_GMRES, restart=20, tol=1e-10, maxit=5000)
    s1 = hip.Solver(M, hip.default_opts(**kw))
    x1, r1 = s1.solve(b)
    s1.destroy()
    sp_ = hip.Solver(M, hip.default_opts(nvirt=nvirt, **kw))
    xp, rp = sp_.solve(b)
    xq, rq = sp_.solve(b)
    sp_.destroy()
    assert rp.status == hip.STATUS_CONVERGED and abs(int(rp.iters) - int(r1.iters)) <= 2
    assert rq.iters == rp.iters and np.array_equal(xp, xq)
    assert np.linalg.norm(xp - x1) / np.linalg.norm(x1) <= 1e-8
    assert np.linalg.norm(b - A @ xp) / np.linalg.norm(b) <= 2e-10


@pytest.mark.gpu
def test_hip_gmres_on_the_raw_reference_matrix(hip, matrix_path, golden_x, golden_meta):
    """The file matrix as-is (unsymmetric by 3.6e-7): GMRES solves A x = b, which
    is NOT CHOLMOD's operator -- the answer differs from the golden vector by
    the documented 6.6e-7, and matches a direct solve of the raw matrix."""
    A = hip.lsbench_matrix_read(matrix_path("xn3b_A_18"))
    n = A.nrows
    b = O.rhs(n)
    s = hip.Solver(A, hip.default_opts(op_mode=hip.OP_RAW, krylov=hip.KRYLOV_GMRES, restart=32,
                                       tol=1e-12, maxit=20000))
    x, res = s.solve(b)
    s.destroy()
    assert res.status == hip.STATUS_CONVERGED
    M = sp.csr_matrix((A.vals, A.cols.astype(np.int64) - 1, A.offs.astype(np.int64)), shape=(n, n))
    xd = sla.spsolve(M.tocsc(), b)
    assert np.linalg.norm(x - xd) / np.linalg.norm(xd) <= 1e-9
    xg = golden_x("xn3b_A_18")
    err = np.linalg.norm(x - xg) / np.linalg.norm(xg)
    assert abs(err - golden_meta["matrices"]["xn3b_A_18"]["raw_vs_S"]) <= 1e-8
    # with the CHOLMOD operator GMRES reaches the golden vector like PCG does
    s = hip.Solver(A, hip.default_opts(krylov=hip.KRYLOV_GMRES, restart=32, tol=1e-12))
    x, res = s.solve(b)
    s.destroy()
    assert res.status == hip.STATUS_CONVERGED
    assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10


@pytest.mark.gpu
def test_hip_gmres_stop_rules(hip):
    A = operators()["convdiff"]
    b = O.rhs(A.shape[0])
    M = _matrix(hip, A)
    s = hip.Solver(M, hip.default_opts(op_mode=hip.OP_RAW, krylov=hip.KRYLOV_GMRES, restart=5,
                                       tol=1e-14, maxit=17))
    x, res = s.solve(b)
    assert res.status == hip.STATUS_MAXIT and res.iters == 17
    xo, ito, _, sto = O.gmres_jacobi(A.indptr, A.indices, A.data, b, 1e-14, 17, 5)
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) <= 1e-8  # same partial solution
    x, res = s.solve(np.zeros_like(b))
    assert res.status == hip.STATUS_CONVERGED and res.iters == 0 and not x.any()
    s.destroy()
