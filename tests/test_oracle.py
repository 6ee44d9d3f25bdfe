"""The oracle pinned: against the reference's own loader (outputs committed in
tests/golden/ref_print, and live against oracle/_ref when it is present),
against exact known answers, and against the golden solutions."""
import ctypes
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLD, ROOT, SPD, TOY
from oracle import oracle as O


def _print_text(A):
    # src/lsbench-csr.c:94-99: "%u %u %lf\n" with row+base, stored col
    rows = np.repeat(np.arange(A.nrows, dtype=np.int64) + A.base, np.diff(A.offs.astype(np.int64)))
    return "".join("%u %u %f\n" % (r, c, v) for r, c, v in zip(rows, A.cols, A.vals)).encode()


LOADER_CASES = ["dup_unsorted_b1", "unsorted_b0", "missing_row_b1", "sci_values_b1"]


@pytest.mark.parametrize("name", TOY + LOADER_CASES)
def test_oracle_loader_matches_reference_printout(name):
    src = os.path.join(GOLD, "matrices" if name in TOY else "loader_cases", name + ".txt")
    want = open(os.path.join(GOLD, "ref_print", name + ".print"), "rb").read()
    assert _print_text(O.matrix_read(src)) == want


@pytest.mark.parametrize("name", SPD)
def test_oracle_loader_matches_reference_md5(name, matrix_path, golden_meta):
    A = O.matrix_read(matrix_path(name))
    m = golden_meta["matrices"][name]
    assert (A.nrows, A.base, A.nnz) == (m["n"], m["base"], m["nnz"])
    assert hashlib.md5(_print_text(A)).hexdigest() == m["print_md5"]


def test_oracle_c_printer_equals_numpy_printer(tmp_path, matrix_path):
    A = O.matrix_read(matrix_path("xn3b_A_18"))
    # the C printer works on the C handle: re-read and print through C
    L = O.lib()
    err = ctypes.create_string_buffer(64)
    h = L.orc_matrix_read(matrix_path("xn3b_A_18").encode(), err, 64)
    out = tmp_path / "p.txt"
    assert L.orc_matrix_print_file(h, str(out).encode()) == 0
    L.orc_csr_free(h)
    assert out.read_bytes() == _print_text(A)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblsbench_ref.so")),
                    reason="oracle/_ref not built (no /root/reference on this machine)")
@pytest.mark.parametrize("name", TOY + ["xn3b_A_18", "tj7a_A_18"] + LOADER_CASES)
def test_oracle_loader_vs_live_reference_library(name, matrix_path):
    """The reference's real lsbench_matrix_read/_print, compiled from its own
    sources (oracle/Makefile `ref`), run here on the same file."""
    src = (os.path.join(GOLD, "loader_cases", name + ".txt") if name in LOADER_CASES
           else matrix_path(name))
    ref = os.path.join(ROOT, "oracle", "_ref", "liblsbench_ref.so")
    code = ("import ctypes;L=ctypes.CDLL(%r);L.lsbench_matrix_read.restype=ctypes.c_void_p;"
            "L.lsbench_matrix_read.argtypes=[ctypes.c_char_p];"
            "L.lsbench_matrix_print.argtypes=[ctypes.c_void_p];"
            "L.lsbench_matrix_print(L.lsbench_matrix_read(%r.encode()));"
            "ctypes.CDLL(None).fflush(None)" % (ref, src))
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True).stdout
    assert out == _print_text(O.matrix_read(src))


@pytest.mark.parametrize("bad,why", [("3 2\n1 1 1\n", "base"), ("0 1\n", "nnz0"),
                                     ("2 1\n1 1 1.0\n2 2 2.0", "entries"),
                                     ("2 1\n1 1 1.0 \n2 2 2.0\n", "entries"),
                                     ("x y\n", "meta")])
def test_oracle_loader_failure_modes(tmp_path, bad, why):
    # src/lsbench-csr.c:38-43,51-52 (a missing final newline is fatal)
    p = tmp_path / "bad.txt"
    p.write_text(bad)
    with pytest.raises(O.OracleError, match=why):
        O.matrix_read(str(p))


def test_operator_upper_c_equals_scipy(matrix_path):
    for name in ["xn3b_A_18", "tj7a_A_12", "A0_02x02", "A1_02x02"]:
        A = O.matrix_read(matrix_path(name))
        S, S2 = O.operator_upper(A), O.operator_upper_numpy(A)
        assert np.array_equal(S.offs, S2.indptr)
        assert np.array_equal(S.cols, S2.indices)
        assert np.array_equal(S.vals, S2.data)
        D = S2.toarray() if A.nrows < 10 else None
        if D is not None:
            assert np.array_equal(D, D.T)


def test_operator_upper_drops_lower_triangle():
    # unsymmetric values: S must take the UPPER ones (src/cholmod-impl.h:13-16)
    A = O.Csr(2, 0, np.array([0, 2, 4], np.uint32), np.array([0, 1, 0, 1], np.uint32),
              np.array([2.0, 7.0, -3.0, 5.0]))
    S = O.operator_upper(A)
    assert S.to_scipy().toarray().tolist() == [[2.0, 7.0], [7.0, 5.0]]


@pytest.mark.parametrize("name,want", [("I1_05x05", [0, 1 / 2, 2 / 3, 3 / 4, 4 / 5]),
                                       ("A0_02x02", [0.5, -0.5]), ("A1_02x02", [0.5, -0.5])])
def test_pcg_known_answers(name, want, matrix_path, golden_x):
    A = O.matrix_read(matrix_path(name))
    S = O.operator_upper(A)
    x, it, rel, st = O.pcg_jacobi(S.offs, S.cols, S.vals, O.rhs(A.nrows), 1e-14)
    assert st == 1 and it <= A.nrows
    assert np.allclose(x, want, rtol=1e-15, atol=1e-16)
    assert np.allclose(golden_x(name), want, rtol=1e-15, atol=1e-16)


@pytest.mark.parametrize("name", SPD)
def test_pcg_reaches_golden(name, matrix_path, golden_x, golden_meta):
    """Stated tolerance of the whole project: ||x - x_golden|| / ||x_golden||
    <= 1e-10 at PCG tol 1e-12."""
    A = O.matrix_read(matrix_path(name))
    S = O.operator_upper(A)
    b = O.rhs(A.nrows)
    x, it, rel, st = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12)
    xg = golden_x(name)
    m = golden_meta["matrices"][name]
    assert st == 1 and it == m["pcg_tol1e-12"]["iters"]
    assert np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
    # the golden vector itself: residual and cross-solver spread
    r = b - O.spmv(S.offs, S.cols, S.vals, xg)
    assert np.linalg.norm(r) / np.linalg.norm(b) <= 1e-12
    assert m["spread_vs_superlu"] <= 5e-14
    # the parity trap: the raw file matrix gives a different answer
    assert m["raw_vs_S"] > 1e-8


def test_pcg_threads_agree(matrix_path):
    A = O.matrix_read(matrix_path("xn3b_A_18"))
    S = O.operator_upper(A)
    b = O.rhs(A.nrows)
    x1, it1, _, _ = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12, threads=1)
    x4, it4, _, _ = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12, threads=4)
    assert abs(it1 - it4) <= 2
    assert np.linalg.norm(x1 - x4) / np.linalg.norm(x1) < 1e-11


def test_spmv_vs_scipy():
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    M = sp.random(500, 500, 0.02, random_state=1, format="csr")
    M.sort_indices()
    x = rng.standard_normal(500)
    y = O.spmv(M.indptr, M.indices, M.data, x)
    assert np.allclose(y, M @ x, rtol=1e-13, atol=1e-13)


def test_laplacians_vs_kron():
    import scipy.sparse as sp

    def t(n):
        return sp.diags([-1, 2, -1], [-1, 0, 1], shape=(n, n))
    nx, ny, nz = 7, 5, 4
    o, c, v = O.lap2d(nx, ny)
    want = sp.kron(sp.eye(ny), t(nx)) + sp.kron(t(ny), sp.eye(nx))
    assert (sp.csr_matrix((v, c, o.astype(np.int64)), shape=(nx * ny,) * 2) != want).nnz == 0
    o, c, v = O.lap3d(nx, ny, nz)
    want = (sp.kron(sp.eye(nz), sp.kron(sp.eye(ny), t(nx))) +
            sp.kron(sp.eye(nz), sp.kron(t(ny), sp.eye(nx))) +
            sp.kron(t(nz), sp.eye(nx * ny)))
    assert (sp.csr_matrix((v, c, o.astype(np.int64)), shape=(nx * ny * nz,) * 2) != want).nnz == 0
    # counts quoted in SURVEY.md section 8(d)
    assert 5 * 3162 * 3162 - 4 * 3162 == 49978572
    assert 7 * 400 ** 3 - 6 * 400 ** 2 == 447040000


def test_powerlaw_definition():
    g = O.powerlaw_gamma(32, 4096)
    thr, mean = O.powerlaw_table(g, 4096)
    assert abs(mean - 32) < 1e-6 and abs(g - 1.585350372615855) < 1e-9
    n = 50000
    o, c, v = O.powerlaw(n, thr, 20240607)
    d = np.diff(o.astype(np.int64))
    assert d.min() >= 1 and d.max() <= 4096 and 25 < d.mean() < 40
    for i in (0, 17, n - 1, int(np.argmax(d))):
        cc = c[o[i]:o[i + 1]].astype(np.int64)
        assert np.all(np.diff(cc) > 0) and cc[0] >= 0 and cc[-1] < n  # sorted, distinct
    assert -1 <= v.min() and v.max() < 1
    # any row range reproduces the same rows
    o2, c2, v2 = O.powerlaw(n, thr, 20240607, 1000, 1500)
    assert np.array_equal(c2, c[o[1000]:o[1500]]) and np.array_equal(v2, v[o[1000]:o[1500]])


def test_l1_jacobi_pcg_oracle(matrix_path, golden_x):
    """M = diag(sum_j |S_ij|).  Same solution as the
    direct solve; on a 5-point Laplacian interior rows have l1 norm 8 = 2 x the
    diagonal, so the iteration count equals plain Jacobi's (M is a multiple of
    it up to the boundary rows) within a few."""
    A = O.matrix_read(matrix_path("xn3b_A_18"))
    S = O.operator_upper(A)
    b = O.rhs(S.nrows)
    x, it, rel, st = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12, jacobi=2)
    xg = golden_x("xn3b_A_18")
    assert st == 1 and np.linalg.norm(x - xg) / np.linalg.norm(xg) <= 1e-10
    # against a numpy statement of the same recurrences
    M = S.to_scipy()
    dinv = 1.0 / np.asarray(abs(M).sum(axis=1)).ravel()
    xr = np.zeros(S.nrows)
    r = b.copy()
    z = dinv * r
    p = z.copy()
    rz = r @ z
    k = 0
    while True:
        q = M @ p
        a = rz / (p @ q)
        xr += a * p
        r -= a * q
        k += 1
        if r @ r <= 1e-24 * (b @ b):
            break
        z = dinv * r
        rz, rz0 = r @ z, rz
        p = z + (rz / rz0) * p
    assert abs(k - it) <= 2 and np.linalg.norm(xr - x) / np.linalg.norm(x) <= 1e-9
    o, c, v = O.lap2d(60, 50)
    _, it1, _, _ = O.pcg_jacobi(o, c, v, O.rhs(3000), 1e-10, jacobi=1)
    _, it2, _, _ = O.pcg_jacobi(o, c, v, O.rhs(3000), 1e-10, jacobi=2)
    assert abs(it1 - it2) <= max(4, it1 // 10)
