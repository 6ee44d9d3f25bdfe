"""Reordering: reverse Cuthill-McKee on the
host + symmetric permutation, with the reference's semantics for a permuted
solve (src/cusparse.c:67-97 permutation, :177 rhs, :204 un-permute)."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

import lsbench_amd as la
from oracle import oracle as O


def _scipy(A):
    return sp.csr_matrix((A.vals, A.cols.astype(np.int64) - A.base, A.offs.astype(np.int64)),
                         shape=(A.nrows, A.nrows))


def _shuffled_laplacian(nx, ny, seed=0):
    A = la.lsbench_matrix_synth("lap2d:nx=%d,ny=%d" % (nx, ny))
    M = _scipy(A)
    q = np.random.default_rng(seed).permutation(A.nrows)
    Ms = M[q][:, q].tocsr()
    Ms.sort_indices()
    return la.Matrix.from_arrays(Ms.indptr, Ms.indices, Ms.data), Ms


def test_rcm_is_a_permutation_and_shrinks_the_band(matrix_path):
    S, Ms = _shuffled_laplacian(60, 45)
    n = S.nrows
    perm = la.lsb_csr_rcm(S)
    assert sorted(perm.tolist()) == list(range(n))
    B = la.lsb_csr_permute_sym(S, perm)
    assert abs(_scipy(B) - Ms[perm][:, perm]).max() == 0          # B = P S P^T exactly
    assert np.all(np.diff(B.offs.astype(np.int64)) == np.diff(Ms.indptr)[perm])
    for i in range(0, n, 97):                                      # rows stay sorted
        assert np.all(np.diff(B.cols[B.offs[i]:B.offs[i + 1]].astype(np.int64)) > 0)
    ps = reverse_cuthill_mckee(Ms, symmetric_mode=True)
    bw_scipy = abs(Ms[ps][:, ps].tocoo().row - Ms[ps][:, ps].tocoo().col).max()
    assert la.lsb_csr_bandwidth(S) > 1000
    assert la.lsb_csr_bandwidth(B) <= 1.25 * bw_scipy + 2          # as good as scipy's RCM
    # reference matrices: already banded, RCM must not make them (much) worse,
    # and a disconnected pattern (diagonal matrix) is handled component by component
    A = la.lsb_csr_symmetrize_upper(la.lsbench_matrix_read(matrix_path("tj7a_A_18")))
    pa = la.lsb_csr_rcm(A)
    assert sorted(pa.tolist()) == list(range(A.nrows))
    assert la.lsb_csr_bandwidth(la.lsb_csr_permute_sym(A, pa)) <= 2 * la.lsb_csr_bandwidth(A)
    D = la.lsb_csr_copy_base0(la.lsbench_matrix_read(matrix_path("I1_05x05")))
    assert sorted(la.lsb_csr_rcm(D).tolist()) == [0, 1, 2, 3, 4]


def test_permuted_system_has_the_permuted_solution():
    S, Ms = _shuffled_laplacian(25, 20, seed=3)
    n = S.nrows
    b = O.rhs(n)
    perm = la.lsb_csr_rcm(S)
    B = la.lsb_csr_permute_sym(S, perm)
    x, it, _, st = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12)
    xp, itp, _, stp = O.pcg_jacobi(B.offs, B.cols, B.vals, b[perm], 1e-12)
    xu = np.empty(n)
    xu[perm] = xp                                                   # src/cusparse.c:204
    assert st == stp == 1 and abs(it - itp) <= 2
    assert np.linalg.norm(xu - x) / np.linalg.norm(x) <= 1e-10


@pytest.mark.gpu
def test_hip_reordered_solve_and_spmv(hip, matrix_path, golden_x):
    import torch
    # the reference matrix: same answer with and without reordering
    A = hip.lsbench_matrix_read(matrix_path("xn3b_A_15"))
    b = O.rhs(A.nrows)
    xg = golden_x("xn3b_A_15")
    s0 = hip.Solver(A, hip.default_opts())
    x0, r0 = s0.solve(b)
    s0.destroy()
    s1 = hip.Solver(A, hip.default_opts(reorder=1))
    x1, r1 = s1.solve(b)
    assert r1.status == hip.STATUS_CONVERGED and abs(int(r1.iters) - int(r0.iters)) <= 2
    assert np.linalg.norm(x1 - xg) / np.linalg.norm(xg) <= 1e-10
    assert np.linalg.norm(x1 - x0) / np.linalg.norm(x0) <= 1e-11
    v = np.random.default_rng(1).standard_normal(A.nrows)
    d_y = torch.empty(A.nrows, dtype=torch.float64, device="cuda:0")
    s1.spmv_dev(torch.from_numpy(v).to("cuda:0"), d_y)             # y = S v in ORIGINAL order
    So = O.operator_upper(O.matrix_read(matrix_path("xn3b_A_15")))
    assert np.allclose(d_y.cpu().numpy(), O.spmv(So.offs, So.cols, So.vals, v), rtol=1e-12, atol=1e-9)
    s1.destroy()
    # a shuffled Laplacian: RCM restores the band; GMRES path too
    S, Ms = _shuffled_laplacian(300, 200, seed=5)
    bs = O.rhs(S.nrows)
    xo, ito, _, _ = O.pcg_jacobi(S.offs, S.cols, S.vals, bs, 1e-10)
    for kry in (hip.KRYLOV_PCG, hip.KRYLOV_GMRES):
        s = hip.Solver(S, hip.default_opts(op_mode=hip.OP_RAW, reorder=1, tol=1e-10, krylov=kry,
                                           maxit=50000))
        x, r = s.solve(bs)
        s.destroy()
        assert r.status == hip.STATUS_CONVERGED
        assert np.linalg.norm(bs - Ms @ x) / np.linalg.norm(bs) <= 1e-9
        if kry == hip.KRYLOV_PCG:
            assert abs(int(r.iters) - ito) <= 3
