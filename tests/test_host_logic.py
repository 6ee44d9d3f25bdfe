"""Host-side C of the product (loader, operator build, partitioner, row blocks,
synthetic generators, CLI) against the oracle and the reference's own
print-outs.  No GPU."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import lsbench_amd as la
from conftest import GOLD, ROOT, SPD, TOY
from oracle import oracle as O

LOADER_CASES = ["dup_unsorted_b1", "unsorted_b0", "missing_row_b1", "sci_values_b1"]
DRIVER = os.path.join(ROOT, "lsbench_amd", "csrc", "driver")


def _print_via_library(path):
    """stdout of the product's lsbench_matrix_print(lsbench_matrix_read(path))."""
    code = ("import sys;sys.path.insert(0,%r);import ctypes;from lsbench_amd import _lib;"
            "L=_lib.load();A=L.lsbench_matrix_read(%r.encode());L.lsbench_matrix_print(A);"
            "ctypes.CDLL(None).fflush(None)" % (ROOT, path))
    return subprocess.run([sys.executable, "-c", code], check=True, capture_output=True).stdout


@pytest.mark.parametrize("name", TOY + LOADER_CASES)
def test_loader_matches_reference_printout(name):
    src = os.path.join(GOLD, "matrices" if name in TOY else "loader_cases", name + ".txt")
    want = open(os.path.join(GOLD, "ref_print", name + ".print"), "rb").read()
    assert _print_via_library(src) == want


@pytest.mark.parametrize("name", SPD)
def test_loader_matches_oracle_and_reference_md5(name, matrix_path, golden_meta):
    p = matrix_path(name)
    A, Ao = la.lsbench_matrix_read(p), O.matrix_read(p)
    assert (A.nrows, A.base) == (Ao.nrows, Ao.base)
    assert np.array_equal(A.offs, Ao.offs) and np.array_equal(A.cols, Ao.cols)
    assert np.array_equal(A.vals, Ao.vals)  # bit-exact: same strtod
    if name in ("xn3b_A_10", "tj7a_A_18"):
        assert hashlib.md5(_print_via_library(p)).hexdigest() == \
            golden_meta["matrices"][name]["print_md5"]


def test_loader_random_files_match_oracle(tmp_path):
    """Unsorted, duplicated (up to 4x), both bases, sparse row ids: both loaders
    must agree bit for bit (duplicates are summed in file order)."""
    rng = np.random.default_rng(3)
    for case in range(6):
        base = case % 2
        n = int(rng.integers(1, 60))
        m = int(rng.integers(1, 400))
        r = rng.integers(0, n, m) + base
        c = rng.integers(0, n, m) + base
        if case >= 4:  # huge row ids => comparison-sort path of the product
            r = r.astype(np.int64) * 50000000 + base
        v = rng.standard_normal(m)
        p = tmp_path / ("m%d.txt" % case)
        with open(p, "w") as f:
            f.write("%d %d\n" % (m, base))
            for i in range(m):
                f.write("%d %d %.17g\n" % (r[i], c[i], v[i]))
        A, Ao = la.lsbench_matrix_read(p), O.matrix_read(str(p))
        assert A.nrows == Ao.nrows and np.array_equal(A.offs, Ao.offs)
        assert np.array_equal(A.cols, Ao.cols) and np.array_equal(A.vals, Ao.vals)


@pytest.mark.parametrize("text,msg", [
    ("3 2\n1 1 1\n", "Base should be either 0 or 1"),
    ("0 1\n", "nnz values in the file are zero"),
    ("2 1\n1 1 1.0\n2 2 2.0", "Unable to read matrix entries"),   # no final newline
    ("2 1\n1 1 1.0 \n2 2 2.0\n", "Unable to read matrix entries"),  # blank before \n
    ("x y\n", "Unable to read meta information"),
    ("3 1\n1 1 1.0\n", "Unable to read matrix entries"),           # fewer records
])
def test_loader_failure_modes_exit_like_reference(tmp_path, text, msg):
    # src/lsbench-csr.c:38-43,51-52: errx(EXIT_FAILURE, ...)
    p = tmp_path / "bad.txt"
    p.write_text(text)
    r = subprocess.run([DRIVER, "--solver", "hip", "--matrix", str(p)], capture_output=True,
                       text=True)
    assert r.returncode == 1 and msg in r.stderr


def test_loader_missing_file(tmp_path):
    r = subprocess.run([DRIVER, "--solver", "hip", "--matrix", str(tmp_path / "nope.txt")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Unable to open file" in r.stderr


@pytest.mark.parametrize("name", TOY + SPD)
def test_symmetrize_upper_equals_oracle(name, matrix_path):
    p = matrix_path(name)
    A, Ao = la.lsbench_matrix_read(p), O.matrix_read(p)
    S, So = la.lsb_csr_symmetrize_upper(A), O.operator_upper(Ao)
    assert S.base == 0 and S.nrows == So.nrows
    assert np.array_equal(S.offs, So.offs) and np.array_equal(S.cols, So.cols)
    assert np.array_equal(S.vals, So.vals)


def test_symmetrize_takes_upper_and_handles_missing_diagonal():
    A = la.Matrix.from_arrays([0, 2, 4, 5], [1, 2, 1, 3, 1], [7.0, 1.0, 9.0, 5.0, 4.0], base=1)
    # rows (1-based): r1: (1,1)=7 (1,2)=1 ; r2: (2,1)=9 (2,3)=5 ; r3: (3,1)=4  -> no diag in r2,r3
    S = la.lsb_csr_symmetrize_upper(A)
    import scipy.sparse as sp
    D = sp.csr_matrix((S.vals, S.cols, S.offs.astype(np.int64)), shape=(3, 3)).toarray()
    assert D.tolist() == [[7, 1, 0], [1, 0, 5], [0, 5, 0]]
    So = O.operator_upper(O.Csr(3, 1, np.array([0, 2, 4, 5], np.uint32),
                                np.array([1, 2, 1, 3, 1], np.uint32),
                                np.array([7.0, 1.0, 9.0, 5.0, 4.0])))
    assert np.array_equal(S.cols, So.cols) and np.array_equal(S.vals, So.vals)


def test_copy_base0_and_row_slice(matrix_path):
    A = la.lsbench_matrix_read(matrix_path("xn3b_A_18"))
    B = la.lsb_csr_copy_base0(A)
    assert B.base == 0 and np.array_equal(B.cols, A.cols - 1) and np.array_equal(B.vals, A.vals)
    R = la.lsb_csr_row_slice(B, 100, 900)
    assert R.nrows == 800 and np.array_equal(R.offs, B.offs[100:901] - B.offs[100])
    assert np.array_equal(R.cols, B.cols[B.offs[100]:B.offs[900]])  # global ids kept
    E = la.lsb_csr_row_slice(B, 5, 5)
    assert E.nrows == 0 and E.nnz == 0


@pytest.mark.parametrize("cap", [1, 7, 64, 2048])
def test_row_blocks_invariants(cap, matrix_path):
    thr, _ = O.powerlaw_table(1.2, 512)
    o, c, v = O.powerlaw(3000, thr, 9)
    mats = [la.Matrix.from_arrays(o, c, v),
            la.lsb_csr_symmetrize_upper(la.lsbench_matrix_read(matrix_path("tj7a_A_18"))),
            # empty rows, a run of them, and an empty tail
            la.Matrix.from_arrays([0, 0, 0, 3, 3, 3, 3, 9, 9], [0, 1, 2, 0, 1, 2, 3, 4, 5],
                                  np.ones(9))]
    for A in mats:
        rb = la.lsb_csr_row_blocks(A, cap).astype(np.int64)
        offs = A.offs.astype(np.int64)
        assert rb[0] == 0 and rb[-1] == A.nrows and np.all(np.diff(rb) >= 1)
        cnt = offs[rb[1:]] - offs[rb[:-1]]
        over = cnt > cap
        assert np.all(np.diff(rb)[over] == 1)  # an oversize block is one long row
        # greedy: a block could not have taken the next row as well
        for k in range(len(rb) - 2):
            assert offs[rb[k + 1] + 1] - offs[rb[k]] > cap


def test_block_lanes():
    thr, _ = O.powerlaw_table(1.2, 3000)
    o, c, v = O.powerlaw(4000, thr, 5)
    for A in (la.Matrix.from_arrays(o, c, v), la.lsbench_matrix_synth("lap2d:nx=50,ny=50")):
        rb = la.lsb_csr_row_blocks(A, 2048).astype(np.int64)
        L = la.lsb_csr_block_lanes(A, rb).astype(np.int64)
        lens = np.diff(A.offs.astype(np.int64))
        for k in range(len(rb) - 1):
            nr, mx = rb[k + 1] - rb[k], lens[rb[k]:rb[k + 1]].max()
            assert L[k] in (1, 2, 4, 8, 16, 32, 64)
            assert L[k] == 64 or L[k] * 16 >= mx            # <= 16 products per lane
            fill = max(l for l in (1, 2, 4, 8, 16, 32, 64) if l == 1 or nr * l <= 256)
            assert L[k] >= fill                             # never fewer than one pass needs
    assert set(L[:-1]) == {1}  # full blocks of 5-point rows: one lane per row


@pytest.mark.parametrize("P", [1, 2, 3, 4, 8])
def test_partition_rows_balanced_and_even(P, matrix_path):
    S = la.lsb_csr_symmetrize_upper(la.lsbench_matrix_read(matrix_path("xn3b_A_10")))
    b = la.lsb_csr_partition_rows(S, P)
    assert b[0] == 0 and b[-1] == S.nrows and np.all(np.diff(b) >= 0)
    assert np.all(b[1:-1] % 2 == 0)
    nnz = np.diff(S.offs.astype(np.int64)[b])
    assert nnz.max() <= S.nnz / P + 2 * 80  # within two rows of perfect


def test_plan_exchange_pairs_up():
    """Every send of shard a to b is the receive b expects from a; a banded
    operator exchanges with neighbours only."""
    A = la.lsbench_matrix_synth("lap2d:nx=40,ny=37")
    P = 5
    b = la.lsb_csr_partition_rows(A, P)
    hull = np.zeros((P, 4), np.uint32)
    for q in range(P):
        lo, hi = la.lsb_csr_col_hull(la.lsb_csr_row_slice(A, int(b[q]), int(b[q + 1])))
        hull[q] = (b[q], b[q + 1] - b[q], lo, hi)
    plans = [la.lsb_plan_exchange(q, hull) for q in range(P)]
    for q, (recv, send) in enumerate(plans):
        assert all(abs(p - q) == 1 for p, _, _ in recv + send)
        assert all(cnt == 40 for _, _, cnt in recv + send)  # one grid line
        for peer, off, cnt in send:
            assert (q, off, cnt) in plans[peer][0]
            assert b[q] <= off and off + cnt <= b[q + 1]
        for peer, off, cnt in recv:
            assert (q, off, cnt) in plans[peer][1]


def test_synth_equals_oracle():
    L = la.lsbench_matrix_synth("lap2d:nx=61,ny=47")
    o, c, v = O.lap2d(61, 47)
    assert L.n_global == 61 * 47 and np.array_equal(L.offs, o)
    assert np.array_equal(L.cols, c) and np.array_equal(L.vals, v)
    L = la.lsbench_matrix_synth("lap3d:nx=13,ny=11,nz=9", 200, 901)
    o, c, v = O.lap3d(13, 11, 9, 200, 901)
    assert L.nrows == 701 and L.n_global == 13 * 11 * 9
    assert np.array_equal(L.offs, o) and np.array_equal(L.cols, c) and np.array_equal(L.vals, v)
    g = 1.585350372615855
    thr, _ = O.powerlaw_table(g, 4096)
    for r0, r1 in [(0, 0), (777, 4001)]:
        L = la.lsbench_matrix_synth("powerlaw:n=12000,gamma=%r,max=4096,seed=20240607" % g, r0, r1)
        o, c, v = O.powerlaw(12000, thr, 20240607, r0, r1 or 12000)
        assert np.array_equal(L.offs, o) and np.array_equal(L.cols, c)
        assert np.array_equal(L.vals, v)
    # avg= resolves to the same exponent as the oracle's bisection (to 1e-9)
    L1 = la.lsbench_matrix_synth("powerlaw:n=3000,avg=32,max=4096,seed=1")
    L2 = la.lsbench_matrix_synth("powerlaw:n=3000,gamma=%r,max=4096,seed=1" % g)
    assert abs(L1.nnz - L2.nnz) <= 0.001 * L2.nnz
    with pytest.raises(la.LsbenchHipError):
        la.lsbench_matrix_synth("hilbert:n=4")


@pytest.mark.parametrize("spec,dims,rng_rows", [
    ("lap2d:nx=61,ny=47,coef=1", (61, 47, None), (0, None)),
    ("lap2d:nx=33,ny=20,coef=77", (33, 20, None), (101, 500)),
    ("lap3d:nx=13,ny=11,nz=9,coef=1", (13, 11, 9), (0, None)),
    ("lap3d:nx=7,ny=6,nz=5,coef=3", (7, 6, 5), (40, 171)),
    ("lap2d:nx=128,ny=1,coef=1", (128, 1, None), (0, None))])
def test_variable_coefficient_operator(spec, dims, rng_rows):
    """`coef=K`: the lap2d / lap3d pattern with general values (one weight per grid
    edge, Dirichlet).  Product generator == oracle's independent statement bit for
    bit, any row range; same pattern as the constant operator; exactly symmetric;
    weights in [1/2, 3/2); row sums = the weights of the edges leaving the grid."""
    import scipy.sparse as sp
    nx, ny, nz = dims
    r0, r1 = rng_rows
    n = nx * ny * (nz or 1)
    L = la.lsbench_matrix_synth(spec, r0, r1 or 0)
    o, c, v = O.lap_coef(nx, ny, nz, int(spec.rsplit("=", 1)[1]), r0, r1)
    assert L.n_global == n and np.array_equal(L.offs, o)
    assert np.array_equal(L.cols, c) and np.array_equal(L.vals, v)
    plain = la.lsbench_matrix_synth(spec.rsplit(",", 1)[0], r0, r1 or 0)
    assert np.array_equal(plain.offs, L.offs) and np.array_equal(plain.cols, L.cols)
    full = la.lsbench_matrix_synth(spec)
    S = sp.csr_matrix((full.vals, full.cols.astype(np.int64), full.offs.astype(np.int64)), shape=(n, n))
    assert (S != S.T).nnz == 0                           # exactly symmetric
    off = S - sp.diags(S.diagonal())
    assert off.data.max() <= -0.5 and off.data.min() > -1.5
    rs = np.asarray(S.sum(axis=1)).ravel()               # = sum of the ghost-edge weights
    assert rs.min() >= -16 * np.finfo(float).eps         # round-off of a row sum of ~9
    i = np.arange(n) % nx
    j = (np.arange(n) // nx) % ny
    k = np.arange(n) // (nx * ny)
    interior = (i > 0) & (i < nx - 1) & (j > 0) & (j < ny - 1)
    if nz:
        interior &= (k > 0) & (k < nz - 1)
    assert np.all(np.abs(rs[interior]) <= 16 * np.finfo(float).eps) and np.all(rs[~interior] >= 0.5 - 1e-15)
    assert len(np.unique(full.vals)) > 0.9 * full.nnz / 2  # general values: nothing to elide
    if n <= 4000:                                        # SPD (dense check on the small ones)
        assert np.linalg.eigvalsh(S.toarray()).min() > 0


def test_synth_through_matrix_read_prefix():
    A = la.lsbench_matrix_read("synth:lap2d:nx=9,ny=4")
    assert A.nrows == 36 and A.nnz == 5 * 36 - 2 * 9 - 2 * 4


def test_full_size_laplacian_counts():
    """BASELINE config 3 at full size on the host: nnz and row sums
    (size-independent properties; the GPU tests use the same generator)."""
    A = la.lsbench_matrix_synth("lap2d:nx=3162,ny=3162")
    assert A.nrows == 9998244 and A.nnz == 49978572
    rowlen = np.diff(A.offs.astype(np.int64))
    assert rowlen.min() == 3 and rowlen.max() == 5
    rs = np.add.reduceat(A.vals, A.offs[:-1].astype(np.int64))
    assert set(np.unique(rs)) == {0.0, 1.0, 2.0}  # 4 - number of neighbours


# ---- CLI (src/lsbench.c:82-150) --------------------------------------------

def _run(*args):
    return subprocess.run([DRIVER] + list(args), capture_output=True, text=True)


def test_cli_help_and_errors():
    r = _run("--help")
    assert r.returncode == 0 and "--matrix" in r.stdout and "hip" in r.stdout
    r = _run("--bogus")
    assert r.returncode == 1
    r = _run("--solver", "hip")
    assert r.returncode == 1 and "Input matrix file not provided" in r.stderr
    m = os.path.join(GOLD, "matrices", "I1_05x05.txt")
    r = _run("--matrix", m, "--precision", "fp32")
    assert r.returncode == 1 and "Precisions other than FP64" in r.stderr


def test_cli_fallbacks_and_disabled_backends():
    m = os.path.join(GOLD, "matrices", "I1_05x05.txt")
    r = _run("--matrix", m, "--solver", "nonsense", "--ordering=zzz", "--trials", "3")
    # unknown solver -> CHOLMOD with a warning (src/lsbench.c:32-33); CHOLMOD is
    # not built here -> the disabled-backend no-op, exit 0
    assert r.returncode == 0
    assert "Invalid solver" in r.stderr and "Invalid ordering" in r.stderr
    assert "not built into this library" in r.stderr
    r = _run("--matrix", m)  # default solver = enum 0 = cusolver: also a no-op
    assert r.returncode == 0 and "===matrix" not in r.stdout


def test_cli_hip_without_gpu_is_quiet_noop():
    if la._lib.load().lsb_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    m = os.path.join(GOLD, "matrices", "I1_05x05.txt")
    r = _run("--matrix", m, "--solver", "hip", "--trials=2")
    assert r.returncode == 0 and "no usable GPU" in r.stderr and "===matrix" not in r.stdout


def _sell_spmv(sptr, cols, vals, x, n):
    """numpy evaluation of the sliced-ELL layout lsb_csr_sellize documents:
    entry j of row 128 s + i sits at sptr[s] + 128 j + i."""
    R = la.SELL_ROWS
    y = np.zeros(len(sptr) * R - R)
    for s in range(len(sptr) - 1):
        ln = (int(sptr[s + 1]) - int(sptr[s])) // R
        blkc = cols[sptr[s]:sptr[s + 1]].reshape(ln, R)
        blkv = vals[sptr[s]:sptr[s + 1]].reshape(ln, R)
        acc = np.zeros(R)
        for j in range(ln):                       # column order, like a CSR row loop
            acc += blkv[j] * x[blkc[j]]
        y[s * R:(s + 1) * R] = acc
    return y[:n]


@pytest.mark.parametrize("spec", ["lap2d:nx=37,ny=23", "lap3d:nx=9,ny=8,nz=7",
                                  "powerlaw:n=1500,avg=6,max=40,seed=4", "lap2d:nx=128,ny=1"])
def test_sliced_ell_copy(spec):
    A = la.lsbench_matrix_synth(spec)
    sptr, cols, vals = la.lsb_csr_sellize(A)
    n, R = A.nrows, la.SELL_ROWS
    assert len(sptr) == (n + R - 1) // R + 1 and sptr[0] == 0
    lens = np.diff(A.offs.astype(np.int64))
    for s in range(len(sptr) - 1):                # every slice padded to its longest row
        assert (int(sptr[s + 1]) - int(sptr[s])) == R * int(lens[s * R:(s + 1) * R].max())
    assert np.count_nonzero(vals) == np.count_nonzero(A.vals)
    assert cols.min() >= 0 and cols.max() < max(n, int(A.cols.max()) + 1)
    x = np.random.default_rng(5).standard_normal(max(n, int(A.cols.max()) + 1))
    y = _sell_spmv(sptr, cols, vals, x, n)
    yo = O.spmv(A.offs, A.cols, A.vals, x)
    assert np.allclose(y, yo, rtol=1e-13, atol=1e-13)
    # padding entries point at a column the row already references
    for r in (0, n // 2, n - 1):
        s, i = divmod(r, R)
        ln = (int(sptr[s + 1]) - int(sptr[s])) // R
        rowc = cols[sptr[s]:sptr[s + 1]].reshape(ln, R)[:, i]
        assert set(rowc.tolist()) == set(A.cols[A.offs[r]:A.offs[r + 1]].tolist())


def test_sliced_ell_copy_base1(matrix_path):
    A = la.lsbench_matrix_read(matrix_path("xn3b_A_10"))          # base-1 file
    S = la.lsb_csr_symmetrize_upper(A)
    sptr, cols, vals = la.lsb_csr_sellize(A)
    x = np.random.default_rng(6).standard_normal(A.nrows)
    yo = O.spmv(A.offs, A.cols - A.base, A.vals, x)
    assert np.allclose(_sell_spmv(sptr, cols, vals, x, A.nrows), yo, rtol=1e-12, atol=1e-12)
    assert S.nrows == A.nrows


def _sell16_spmv(sptr, codes, sbase, vals, x, n, row_begin=0):
    """numpy evaluation of the 16-bit sliced-ELL layout: the entry in slot j of
    row r has column r + row_begin + base + code, {base, k} = sbase[sptr[s]/128 + j];
    k >= 0: the slot's codes are codes[128 k ..), k = -1: code 0 for the whole
    slot; value 0 = padding (no gather)."""
    R = la.SELL_ROWS
    y = np.zeros((len(sptr) - 1) * R)
    rows = np.arange(R)
    for s in range(len(sptr) - 1):
        ln = (int(sptr[s + 1]) - int(sptr[s])) // R
        v = vals[sptr[s]:sptr[s + 1]].reshape(ln, R)
        bk = sbase[int(sptr[s]) // R:int(sptr[s]) // R + ln].astype(np.int64)
        acc = np.zeros(R)
        for j in range(ln):
            k = int(bk[j, 1])
            cj = codes[k * R:(k + 1) * R].astype(np.int64) if k >= 0 else np.zeros(R, np.int64)
            col = s * R + rows + row_begin + bk[j, 0] + cj
            live = v[j] != 0.0
            assert np.all((col[live] >= 0) & (col[live] < len(x)))
            acc += np.where(live, v[j] * x[np.where(live, col, 0)], 0.0)
        y[s * R:(s + 1) * R] = acc
    return y[:n]


@pytest.mark.parametrize("spec", ["lap2d:nx=37,ny=23", "lap3d:nx=9,ny=8,nz=7", "lap2d:nx=129,ny=1",
                                  "lap3d:nx=200,ny=190,nz=3",         # plane 38000 > 32767: aligned slots
                                  "powerlaw:n=1500,avg=6,max=40,seed=4"])
def test_sliced_ell_copy_16bit(spec):
    A = la.lsbench_matrix_synth(spec)
    out = la.lsb_csr_sellize16(A)
    assert out is not None
    sptr, codes, sbase, vals = out
    n, R = A.nrows, la.SELL_ROWS
    assert len(sptr) == (n + R - 1) // R + 1 and sbase.shape == (sptr[-1] // R, 2)
    kept = sbase[:, 1][sbase[:, 1] >= 0]
    assert np.array_equal(kept, np.arange(len(kept))) and len(codes) == len(kept) * R
    if spec.startswith("lap"):                 # structured grid: no slot needs a code array
        assert len(kept) == 0
    assert np.count_nonzero(vals) == np.count_nonzero(A.vals)
    x = np.random.default_rng(7).standard_normal(n)
    yo = O.spmv(A.offs, A.cols, A.vals, x)
    assert np.allclose(_sell16_spmv(sptr, codes, sbase, vals, x, n), yo, rtol=1e-13, atol=1e-13)
    if spec.startswith("lap3d:nx=200"):
        # +-38000 neighbours cannot share a slot with the +-200 ones: a slice in the
        # middle plane has 7 slots (6 in the outer planes) and boundary rows keep gaps
        # instead of shifting their entries left
        assert set(np.diff(sptr.astype(np.int64)) // R) <= {5, 6, 7}
        assert sptr[-1] > A.nnz
    if spec.startswith("lap2d:nx=37"):
        assert sptr[-1] == la.lsb_csr_sellize(A)[0][-1]          # no extra padding when all deltas fit


def test_sliced_ell_copy_16bit_row_slice_and_refusal():
    spec = "lap3d:nx=40,ny=30,nz=20"
    n = la.lsbench_matrix_synth(spec, 0, 1).n_global
    r0, r1 = 7000, 19000
    A = la.lsbench_matrix_synth(spec, r0, r1)                     # global column ids
    full = la.lsbench_matrix_synth(spec)
    sptr, codes, sbase, vals = la.lsb_csr_sellize16(A, r0)
    x = np.random.default_rng(8).standard_normal(n)
    yo = O.spmv(full.offs, full.cols, full.vals, x)[r0:r1]
    assert np.allclose(_sell16_spmv(sptr, codes, sbase, vals, x, r1 - r0, r0), yo, rtol=1e-13, atol=1e-13)
    # > 255 diagonal bands in one slice: refused (the caller keeps 32-bit columns)
    cols = (np.arange(300, dtype=np.uint32) * 70001)             # 300 bands, 70001 apart
    B = la.Matrix.from_arrays(np.array([0, len(cols)]), cols, np.ones(len(cols)))
    assert la.lsb_csr_sellize16(B) is None
    ok = la.lsb_csr_sellize16(la.Matrix.from_arrays(np.array([0, 255]), cols[:255], np.ones(255)))
    assert ok is not None and ok[0][-1] == 255 * la.SELL_ROWS


def test_binary_matrix_cache(tmp_path, matrix_path, monkeypatch):
    """LSBENCH_MATRIX_CACHE: the parsed CSR is kept
    next to the text; a second read comes from it, a changed text file
    invalidates it, a damaged cache is ignored."""
    import shutil
    src = tmp_path / "m.txt"
    shutil.copy(matrix_path("xn3b_A_10"), src)
    plain = la.lsbench_matrix_read(str(src))
    monkeypatch.setenv("LSBENCH_MATRIX_CACHE", "1")
    first = la.lsbench_matrix_read(str(src))
    cache = tmp_path / "m.txt.lsbcsr"
    assert cache.exists() and cache.stat().st_size > plain.nnz * 12
    second = la.lsbench_matrix_read(str(src))
    for M in (first, second):
        assert (M.nrows, M.base) == (plain.nrows, plain.base)
        assert np.array_equal(M.offs, plain.offs) and np.array_equal(M.cols, plain.cols)
        assert np.array_equal(M.vals, plain.vals)
    # proof that the second read did not parse the text: same size and mtime, other content
    st = src.stat()
    txt = src.read_bytes()
    src.write_bytes(txt)
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns))
    raw = bytearray(cache.read_bytes())
    raw[-8:] = np.float64(12345.0).tobytes()                     # last value of the cache
    cache.write_bytes(bytes(raw))
    assert la.lsbench_matrix_read(str(src)).vals[-1] == 12345.0
    # a newer text file: the cache is stale, the text is parsed again and the cache rewritten
    os.utime(src, ns=(st.st_atime_ns, st.st_mtime_ns + 5_000_000_000))
    fresh = la.lsbench_matrix_read(str(src))
    assert fresh.vals[-1] == plain.vals[-1]
    assert np.frombuffer(cache.read_bytes()[-8:], np.float64)[0] == plain.vals[-1]
    # truncated cache: ignored
    cache.write_bytes(cache.read_bytes()[:200])
    again = la.lsbench_matrix_read(str(src))
    assert np.array_equal(again.vals, plain.vals)
    # right size, magic and source stamp, but offsets that go backwards / a column
    # below the base: rejected like a truncated file, the text is parsed
    good = cache.read_bytes()
    hdr = 8 + 8 + 8 + 4 + 4 + 8
    for pos, val in ((hdr + 4 * 7, np.uint32(plain.offs[9])),          # offs[7] > offs[8]
                     (hdr + 4 * (plain.nrows + 1) + 4 * 3, np.uint32(0))):  # a column 0 in a base-1 matrix
        raw = bytearray(good)
        raw[pos:pos + 4] = val.tobytes()
        cache.write_bytes(bytes(raw))
        got = la.lsbench_matrix_read(str(src))
        assert np.array_equal(got.offs, plain.offs) and np.array_equal(got.cols, plain.cols)
    # a cache directory
    d = tmp_path / "cache"
    d.mkdir()
    monkeypatch.setenv("LSBENCH_MATRIX_CACHE", str(d))
    la.lsbench_matrix_read(str(src))
    assert (d / "m.txt.lsbcsr").exists()


@pytest.mark.parametrize("spec,width", [("powerlaw:n=20000,gamma=1.2,max=3000,seed=7", 1024),
                                        ("powerlaw:n=5000,gamma=1.585350372615855,max=4096,seed=3", 300),
                                        ("lap2d:nx=70,ny=50", 512)])
def test_binned_form(spec, width):
    """lsb_csr_binize (LSB_SPMV_BINNED): a permutation of the entries, bin-major,
    rows ascending inside a bin, column order of a row kept; chunks hold whole
    (row, bin) runs, at most BIN_CHUNK entries unless one run alone is longer;
    y = sum over bins reproduces A x."""
    import ctypes as C
    L = la._lib
    A = la.lsbench_matrix_synth(spec)
    B = L.load().lsb_csr_binize(A.ptr, width).contents
    nnz = int(A.nnz)
    assert (B.nnz, B.nrows, B.width) == (nnz, A.nrows, width)
    rows = np.ctypeslib.as_array(B.rows, (nnz,)).copy()
    cols = np.ctypeslib.as_array(B.cols, (nnz,)).copy()
    vals = np.ctypeslib.as_array(B.vals, (nnz,)).copy()
    cb = np.ctypeslib.as_array(B.chunk_begin, (B.nchunks + 1,)).copy()
    bc = np.ctypeslib.as_array(B.bin_chunk, (B.nbins + 1,)).copy()
    nbins, nchunks = int(B.nbins), int(B.nchunks)
    L.load().lsb_binned_free(C.pointer(B))
    offs = A.offs.astype(np.int64)
    arow = np.repeat(np.arange(A.nrows), np.diff(offs))
    # the same multiset of (row, col, value)
    o1 = np.lexsort((A.cols, arow))
    o2 = np.lexsort((cols, rows))
    assert np.array_equal(arow[o1], rows[o2]) and np.array_equal(A.cols[o1], cols[o2])
    assert np.array_equal(A.vals[o1], vals[o2])
    bins = cols // width
    assert np.all(np.diff(bins) >= 0)                                   # bin-major
    same_bin = np.diff(bins) == 0
    assert np.all(np.diff(rows)[same_bin] >= 0)                         # rows ascend inside a bin
    same_run = same_bin & (np.diff(rows) == 0)
    assert np.all(np.diff(cols.astype(np.int64))[same_run] > 0)         # column order kept
    assert cb[0] == 0 and cb[-1] == nnz and np.all(np.diff(cb) > 0) and bc[0] == 0 and bc[-1] == nchunks
    for b in range(nbins):                                             # chunks do not straddle bins
        e0, e1 = cb[bc[b]], cb[bc[b + 1]]
        assert e0 == e1 or (bins[e0] == b and bins[e1 - 1] == b)
    for k in range(nchunks):
        e0, e1 = cb[k], cb[k + 1]
        if e0 > 0 and bins[e0 - 1] == bins[e0]:
            assert rows[e0 - 1] != rows[e0]                            # boundary = run boundary
        if e1 - e0 > L.BIN_CHUNK:
            assert np.all(rows[e0:e1] == rows[e0])                     # one long run on its own
    x = np.sin(np.arange(A.nrows, dtype=np.float64))
    y = np.zeros(A.nrows)
    np.add.at(y, rows, vals * x[cols])
    assert np.allclose(y, O.spmv(A.offs, A.cols, A.vals, x), rtol=1e-12, atol=1e-12)


def test_powerlaw_spd_variant():
    """`powerlaw:...,spd=1` (SURVEY.md section 8(d), config 5 for CG runs):
    S = (B + B^T) + diag(1 + sum_j |(B + B^T)_ij|) against a scipy construction
    from the ORACLE's generator; any row range equals the rows of the whole."""
    import scipy.sparse as sp
    n, dmax, seed = 3000, 700, 11
    gamma = 1.3
    thr, _ = O.powerlaw_table(gamma, dmax)
    o, c, v = O.powerlaw(n, thr, seed)
    B = sp.csr_matrix((v, c, o.astype(np.int64)), shape=(n, n))
    T = (B + B.T).tocsr()
    T.sort_indices()
    Sref = (T + sp.diags(1.0 + np.asarray(abs(T).sum(axis=1)).ravel())).tocsr()
    Sref.sort_indices()
    spec = "powerlaw:n=%d,gamma=%r,max=%d,seed=%d,spd=1" % (n, gamma, dmax, seed)
    S = la.lsbench_matrix_synth(spec)
    M = sp.csr_matrix((S.vals, S.cols.astype(np.int64), S.offs.astype(np.int64)), shape=(n, n))
    assert M.has_sorted_indices and abs(M - M.T).max() == 0.0           # exactly symmetric
    D = (M - Sref)
    assert abs(D).max() <= 1e-12 * abs(Sref).max()
    assert set(zip(*M.nonzero())) >= set(zip(*Sref.nonzero()))
    d = M.diagonal()
    assert np.all(d >= 1.0 + (abs(M).sum(axis=1).A1 - np.abs(d)) - 1e-9)  # strictly dominant
    part = la.lsbench_matrix_synth(spec, 1000, 1800)
    assert part.n_global == n and part.nrows == 800
    assert np.array_equal(part.offs, S.offs[1000:1801] - S.offs[1000])
    assert np.array_equal(part.cols, S.cols[S.offs[1000]:S.offs[1800]])
    assert np.array_equal(part.vals, S.vals[S.offs[1000]:S.offs[1800]])


@pytest.mark.parametrize("tiling", [(0, 0), (256, 128), (8192, 2048)])
@pytest.mark.parametrize("spec", ["powerlaw:n=30000,gamma=1.2,max=3000,seed=7",
                                  "powerlaw:n=5000,gamma=1.585350372615855,max=4096,seed=3",
                                  "lap2d:nx=170,ny=150", "powerlaw:n=2500,gamma=0.3,max=2500,seed=5"])
def test_twophase_form(spec, tiling):
    """lsb_csr_pbize2 (LSB_SPMV_TWOPHASE): entries ordered by (column chunk, row bin,
    row, column), every chunk starting on a multiple of 64; the product array is row-bin
    major and a PIECE -- the entries of one (chunk, bin) pair -- is contiguous and in the
    same order on both sides, so slot = entry + delta[piece]; the piece of an entry is
    read off grp_first / grp_mask of its group of 64 (what phase 1 does with two scalar
    loads and a popcount); phase-1 items tile the chunks; products summed slot by slot
    give A x."""
    import ctypes as C
    L = la._lib
    A = la.lsbench_matrix_synth(spec)
    P = L.load().lsb_csr_pbize2(A.ptr, tiling[0], tiling[1]).contents
    Cc, R = int(P.cols), int(P.rows)
    assert (Cc, R) == (tiling[0] or L.PB_COLS, tiling[1] or L.PB_ROWS)
    nnz, nent, nb, nch, ni, npc = int(P.nnz), int(P.nent), int(P.nbins), int(P.nchunks), int(P.nitems), int(P.npieces)
    assert nnz == A.nnz and P.nrows == A.nrows and nb == (A.nrows + R - 1) // R and nent % 64 == 0
    vals = np.ctypeslib.as_array(P.vals, (nent,)).copy()
    colw = np.ctypeslib.as_array(P.colw, (nent,)).astype(np.int64)
    first = np.ctypeslib.as_array(P.grp_first, (nent // 64,)).astype(np.int64)
    mask = np.ctypeslib.as_array(P.grp_mask, (nent // 64,)).copy()
    delta = np.ctypeslib.as_array(P.delta, (npc,)).astype(np.int64)
    item = np.ctypeslib.as_array(P.item, (3 * ni,)).reshape(ni, 3).astype(np.int64)
    binptr = np.ctypeslib.as_array(P.bin_ptr, (nb + 1,)).astype(np.int64)
    roww = np.ctypeslib.as_array(P.roww, (nnz,)).astype(np.int64)
    col_lo = int(P.ncols_lo)
    L.load().lsb_pb_free(C.pointer(P))
    assert col_lo % Cc == 0 and colw.max() < Cc
    # items: slices of a chunk, on multiples of 64, chunk after chunk; the gaps hold zeros
    assert item[0, 1] == 0 and np.all(item[:, 1] % 64 == 0) and np.all(np.diff(item[:, 0]) >= 0)
    assert np.all(item[:, 2] - item[:, 1] <= 8192) and np.all(item[1:, 1] - item[:-1, 2] < 64)
    real = np.zeros(nent, bool)
    chunk_of = np.zeros(nent, np.int64)
    for c, e0, e1 in item:
        real[e0:e1] = True
        chunk_of[e0:e1] = c
    assert real.sum() == nnz and np.all(vals[~real] == 0.0) and -(-item[-1, 2] // 64) * 64 == nent
    # the piece of an entry, the way phase 1 computes it
    e = np.nonzero(real)[0]
    lane = e % 64
    le = np.where(lane == 63, np.uint64(0xFFFFFFFFFFFFFFFF), (np.uint64(2) << lane.astype(np.uint64)) - np.uint64(1))
    bits = mask[e // 64] & le
    pop = np.array([bin(int(v)).count("1") for v in bits]) if len(bits) < 400000 else \
        np.unpackbits(bits.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1)
    piece = first[e // 64] + np.asarray(pop, dtype=np.int64)
    assert piece.max() == npc - 1 and np.all(np.diff(piece) >= 0) and np.all(np.diff(piece) <= 1)
    slot = (e + delta[piece]) % (1 << 32)
    assert len(np.unique(slot)) == nnz and slot.max() == nnz - 1       # a permutation of the slots
    assert binptr[0] == 0 and binptr[-1] == nnz and np.all(np.diff(binptr) >= 0) and roww.max() < R
    bin_of_slot = np.searchsorted(binptr, np.arange(nnz), side="right") - 1
    rows = bin_of_slot[slot] * R + roww[slot]
    cols = col_lo + chunk_of[e] * Cc + colw[e]
    # one (chunk, bin) pair per piece, contiguous on both sides
    assert np.all(np.diff(slot)[np.diff(piece) == 0] == 1)
    pair = chunk_of[e] * nb + rows // R
    assert np.all(np.diff(pair) >= 0) and np.array_equal(np.diff(pair) > 0, np.diff(piece) == 1)
    # the same multiset of (row, col, value)
    offs = A.offs.astype(np.int64)
    arow = np.repeat(np.arange(A.nrows), np.diff(offs))
    o1 = np.lexsort((A.cols, arow))
    o2 = np.lexsort((cols, rows))
    assert np.array_equal(arow[o1], rows[o2]) and np.array_equal(A.cols[o1].astype(np.int64), cols[o2])
    assert np.array_equal(A.vals[o1], vals[e][o2])
    x = np.sin(np.arange(max(A.nrows, int(cols.max()) + 1), dtype=np.float64))
    prod = np.zeros(nnz)
    prod[slot] = vals[e] * x[cols]
    y = np.zeros(A.nrows)
    np.add.at(y, bin_of_slot * R + roww, prod)
    assert np.allclose(y, O.spmv(A.offs, A.cols, A.vals, x[:A.nrows]), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("spec,frac_kept", [("lap2d:nx=300,ny=170", 0.5), ("lap3d:nx=40,ny=36,nz=30", 0.6),
                                            ("powerlaw:n=3000,gamma=2.2,max=64,seed=5", 1.0)])
def test_sell16_value_slots(spec, frac_kept):
    """lsb_sell16_value_slots: slots of the 16-bit sliced-ELL copy whose 128 values are one
    non-zero number keep it once; the others keep their values, packed in slot order; base
    and code slot are carried over.  Lossless: every value is recovered bit for bit."""
    import ctypes as C
    lib = la._lib.load()
    A = la.lsbench_matrix_synth(spec)
    p = lib.lsb_csr_sellize16(A.ptr, 0)
    assert p
    S = p.contents
    nq = int(S.stored) // la._lib.SELL_ROWS
    vals = np.ctypeslib.as_array(S.vals, (int(S.stored),)).copy()
    sbase = np.ctypeslib.as_array(S.sbase, (2 * nq,)).copy().reshape(nq, 2)
    v = lib.lsb_sell16_value_slots(p)
    V = v.contents
    assert int(V.nslots) == nq
    slots = np.ctypeslib.as_array(V.slots, (4 * nq,)).copy().reshape(nq, 4)
    vconst = np.ctypeslib.as_array(V.vconst, (nq,)).copy()
    packed = np.ctypeslib.as_array(V.vals, (max(int(V.nval_slots), 1) * 128,)).copy()
    nv = int(V.nval_slots)
    lib.lsb_sell_vc_free(v)
    lib.lsb_sell_free(p)
    assert np.array_equal(slots[:, :2], sbase) and np.all(slots[:, 3] == 0)
    kept = slots[:, 2] >= 0
    assert kept.sum() == nv and np.array_equal(slots[kept, 2], np.arange(nv))
    rebuilt = np.where(kept[:, None], packed.reshape(-1, 128)[np.maximum(slots[:, 2], 0)], vconst[:, None])
    assert np.array_equal(rebuilt.view(np.uint64), vals.reshape(nq, 128).view(np.uint64))
    const = vals.reshape(nq, 128)
    assert np.array_equal(~kept, (const[:, :1] == const).all(axis=1) & (const[:, 0] != 0.0))
    assert nv <= frac_kept * nq         # a stencil: most slots are one number (random values: none)


def test_templates_cover_a_structured_grid():
    """lsb_sell16_templates on the host: a structured grid has a handful of templates, nearly all
    slices share a shaped one [far][c-1, c, c+1][far]; where a grid line ends inside a slice the
    +-1 slots are MASKED constants (the number in the template, 128 bits per slice and slot);
    every template stands exactly for the slot records, constants and values of its slices;
    general values have none."""
    hip = la
    lib = hip._lib.load()
    for spec, nfar in (("lap2d:nx=411,ny=203", 1), ("lap3d:nx=64,ny=64,nz=40", 2), ("lap2d:nx=9000,ny=1", 0)):
        A = hip.lsbench_matrix_synth(spec)
        H = lib.lsb_csr_sellize16(A.ptr, 0)
        V = lib.lsb_sell16_value_slots(H)
        T = lib.lsb_sell16_templates(H, V)
        assert T, spec
        t = T.contents
        assert t.nfar == nfar and t.ntmpl <= 32 and t.nslice == H.contents.nslice
        assert t.shaped * 4 >= t.nslice * 3 and t.covered >= t.shaped
        tid = np.ctypeslib.as_array(t.tid, (t.nslice,))
        assert tid.max() == 255 or t.covered == t.nslice
        assert int((tid != 255).sum()) == t.covered
        for k in set(int(v) for v in np.unique(tid) if v != 255):
            q = t.t[k]
            special = [j for j in range(q.nslots) if q.kind[j] != 0]
            if q.shaped:
                c = nfar + 1
                assert q.nslots == 2 * nfar + 3 and q.base[c - 1] + 1 == q.base[c] == q.base[c + 1] - 1
                assert set(special) <= {c - 1, c + 1}
            else:                                                  # all-gathered templates keep nothing
                assert not special
            assert all((q.kidx[j] >= 0) == (q.kind[j] != 0) for j in range(q.nslots))
        vb = np.ctypeslib.as_array(t.vbase, (2 * t.nslice,)).reshape(-1, 2)
        mask = np.ctypeslib.as_array(t.mask, (2 * max(int(t.nmask), 1),))
        sp = np.ctypeslib.as_array(H.contents.sptr, (t.nslice + 1,)) // 128
        rec = np.ctypeslib.as_array(V.contents.slots, (V.contents.nslots * 4,)).reshape(-1, 4)
        vals = np.ctypeslib.as_array(V.contents.vals, ((V.contents.nval_slots + 1) * 128,)).reshape(-1, 128)
        nmask = kept = 0
        for sl in range(t.nslice):                                 # the records a template stands for
            r = rec[sp[sl]:sp[sl + 1]]
            if tid[sl] == 255:
                kept += int((r[:, 2] >= 0).sum())
                continue
            q = t.t[int(tid[sl])]
            assert q.nslots == len(r) and list(q.base)[:len(r)] == r[:, 0].tolist() and np.all(r[:, 1] < 0)
            for j in range(len(r)):
                if q.kind[j] == 0:
                    assert r[j, 2] < 0 and q.cst[j] == V.contents.vconst[sp[sl] + j]
                elif q.kind[j] == 1:
                    assert vb[sl, 0] + q.kidx[j] == r[j, 2]
                    kept += 1
                else:
                    v = vals[r[j, 2]]
                    m = mask[2 * (vb[sl, 1] + q.kidx[j]):2 * (vb[sl, 1] + q.kidx[j]) + 2]
                    bits = np.array([(int(m[i // 64]) >> (i % 64)) & 1 for i in range(128)], bool)
                    assert np.array_equal(bits, v != 0) and np.all(v[bits] == q.cst[j]) and bits.any()
                    nmask += 1
        assert nmask <= t.nmask and kept == t.kept_read   # (a slice sent back to the per-slot way leaves its masks behind)
        if "lap" in spec and nfar:
            assert t.nmask > 0 and t.kept_read < V.contents.nval_slots    # constant coefficients: masks, few values
        lib.lsb_sell_tmpls_free(T), lib.lsb_sell_vc_free(V), lib.lsb_sell_free(H)
    A = hip.lsbench_matrix_synth("lap2d:nx=411,ny=203,coef=1")
    H = lib.lsb_csr_sellize16(A.ptr, 0)
    V = lib.lsb_sell16_value_slots(H)
    assert not lib.lsb_sell16_templates(H, V)
    lib.lsb_sell_vc_free(V), lib.lsb_sell_free(H)


def test_tmpl_check_states_the_template_kernels_bounds():
    """lsb_tmpl_check (run by the backend at every upload): every unguarded 16-byte gather of
    k_spmv_sell16's constant-slot path and of k_spmv_tmpl stays inside the gather vector, every
    value-slot / mask / template index inside its array -- on whole operators and on shards cut
    inside a plane (row_begin > 0, column ids global).  And the access behind round 3's GPU memory
    fault (gpurun_out/r3_probe18: a template's far gathers issued for a slice that has no template,
    x[row - nx] for rows < nx) is exactly what rule 7 refuses: a constant far slot put on the first
    slice, or a gather vector that starts later than the operator's column 0."""
    import ctypes as C
    lib = la._lib.load()
    why = C.create_string_buffer(256)
    for spec in ("lap2d:nx=411,ny=203", "lap3d:nx=64,ny=64,nz=40", "lap3d:nx=40,ny=36,nz=30", "lap2d:nx=9000,ny=1",
                 "lap2d:nx=411,ny=203,coef=1"):
        n = la.lsbench_matrix_synth(spec, 0, 1).n_global
        for r0, r1 in ((0, n), (n // 3 & ~1, 2 * n // 3 & ~1), (2 * n // 3 & ~1, n)):
            A = la.lsbench_matrix_synth(spec, r0, r1)
            H = lib.lsb_csr_sellize16(A.ptr, r0)
            V = lib.lsb_sell16_value_slots(H)
            T = lib.lsb_sell16_templates(H, V)                     # NULL for general values / small shards
            for deep in (0, 1):
                assert lib.lsb_tmpl_check(H, V, T, r0, A.nrows, n, deep, why, 256) == 0, (spec, r0, why.value)
            # one entry less than the columns the shard references: refused wherever a CONSTANT slot
            # reaches the last column it may (rule 7), or a kept value does (rule 12, deep)
            hi = int(A.cols.max()) + 1
            assert lib.lsb_tmpl_check(H, V, T, r0, A.nrows, hi - 1, 1, why, 256) in (7, 12), (spec, r0)
            if T and r0 == 0:
                t, v = T.contents, V.contents
                # the first slice of the operator has no template (its -nx slot is padding) ...
                assert t.tid[0] == 255 or "ny=1" in spec
                # ... and a constant slot reaching below column 0 there is refused
                q0 = H.contents.sptr[0] // 128
                keep = (v.slots[4 * q0], v.slots[4 * q0 + 2], v.vconst[q0])
                v.slots[4 * q0], v.slots[4 * q0 + 2], v.vconst[q0] = -5, -1, -1.0
                rc = lib.lsb_tmpl_check(H, V, T, r0, A.nrows, n, 0, why, 256)
                v.slots[4 * q0], v.slots[4 * q0 + 2], v.vconst[q0] = keep
                assert rc == 7 and b"gathers x[-5" in why.value
            if T:                                                  # a mask / value index one past its array
                t = T.contents
                if t.nmask:
                    keep, t.nmask = t.nmask, 0
                    assert lib.lsb_tmpl_check(H, V, T, r0, A.nrows, n, 0, why, 256) == 21
                    t.nmask = keep
                lib.lsb_sell_tmpls_free(T)
            lib.lsb_sell_vc_free(V), lib.lsb_sell_free(H)


def test_z_column_plan_of_a_3d_stencil():
    """lsb_sell_tmpl_columns (host side of k_spmv_tmpl_col): on a 3-D stencil whose planes are whole
    slices every slice lands in exactly one item; interior z-columns become runs of 2..kmax slices
    one plane apart that share a template (outermost far slots exactly one plane away, constant or
    masked slots only) and their mask words; the first and last plane -- no neighbour plane on one
    side -- stay single items; the items run z-group after z-group, positions ascending, and every XCD
    takes a contiguous run of them with an equal share of the slices; shards cut inside a plane and
    ragged z-groups included.  lsb_tmpl_cols_check states the rules and catches a broken one."""
    import ctypes as C
    lib = la._lib.load()
    why = C.create_string_buffer(256)
    for spec, period in (("lap3d:nx=128,ny=64,nz=21", 64), ("lap3d:nx=256,ny=32,nz=40", 64), ("lap3d:nx=200,ny=64,nz=30", 100)):
        n = la.lsbench_matrix_synth(spec, 0, 1).n_global
        for r0, r1 in ((0, n), (0, (n // 2 + 4096 + 640) & ~127), (5 * period * 128, n)):
            A = la.lsbench_matrix_synth(spec, r0, r1)
            H = lib.lsb_csr_sellize16(A.ptr, r0)
            V = lib.lsb_sell16_value_slots(H)
            T = lib.lsb_sell16_templates(H, V)
            assert T and T.contents.nfar == 2
            ns = T.contents.nslice
            tid = np.ctypeslib.as_array(T.contents.tid, (ns,))
            for kmax in (2, 5, 8, 16):
                Cp = lib.lsb_sell_tmpl_columns(T, period, kmax)
                if kmax == 2 and not Cp:                               # (pairs on a short shard: under 3/4 in columns)
                    continue
                assert Cp, (spec, r0, kmax)
                c = Cp.contents
                assert lib.lsb_tmpl_cols_check(T, Cp, why, 256) == 0, why.value
                assert c.kmax == kmax and c.period == period and c.centre0 == 1
                it = np.ctypeslib.as_array(c.item, (4 * c.nitem,)).reshape(-1, 4).copy()
                xb = list(c.xbeg)
                assert xb[0] == 0 and xb[8] == c.nitem and xb == sorted(xb)
                seen = np.zeros(ns, int)
                npl = -(-ns // period)
                ng = -(-npl // kmax)
                grp = lambda z: next(g for g in range(ng) if npl * g // ng <= z < npl * (g + 1) // ng)
                cells = [grp(int(s0) // period) * period + int(s0) % period for s0 in it[:, 0]]
                assert cells == sorted(cells)                          # z-group-major, positions ascending
                per_xcd = [int((it[xb[k]:xb[k + 1], 1] & 0x7fffffff).sum()) for k in range(8)]
                assert sum(per_xcd) == ns and max(per_xcd) - min(per_xcd) <= 2 * kmax   # a contiguous, equal share each
                for k in range(8):
                    turn = it[xb[k]:xb[k + 1]]
                    for g in range(0, len(turn), 4):                   # lockstep: a full turn of equal columns
                        runs = turn[g:g + 4, 1]
                        lock = runs >> 31
                        assert lock.min() == lock.max()
                        assert bool(lock[0]) == (len(runs) == 4 and len(set(runs.tolist())) == 1 and (runs[0] & 0x7fffffff) >= 2)
                    for s0, run, t, mb in turn:
                        run &= 0x7fffffff
                        assert 1 <= run <= kmax
                        sl = s0 + period * np.arange(run)
                        seen[sl] += 1
                        if run > 1:
                            assert np.all(tid[sl] == t) and t != 255
                            # a column never leaves its z-group (at most kmax planes, as equal as they come)
                            assert grp(s0 // period) == grp(sl[-1] // period)
                assert np.all(seen == 1)
                runs_all = it[:, 1] & 0x7fffffff
                cols = int(runs_all[runs_all > 1].sum())
                assert cols == c.in_cols and 4 * cols >= 3 * ns
                if r0 == 0:                                            # the operator's first plane: single items
                    assert np.all(runs_all[np.isin(it[:, 0], np.arange(period))] == 1)
                if kmax == 8:                                          # a broken rule is caught
                    big = int(np.argmax(runs_all))
                    keep = int(c.item[4 * big + 1])
                    c.item[4 * big + 1] = (keep & 0x7fffffff) + 1 if (keep & 0x7fffffff) < kmax else (keep & 0x7fffffff) - 1
                    assert lib.lsb_tmpl_cols_check(T, Cp, why, 256) in (5, 6, 7, 9, 12)
                    c.item[4 * big + 1] = keep
                lib.lsb_tmpl_cols_free(Cp)
            assert not lib.lsb_sell_tmpl_columns(T, period + 1, 8)                  # not this operator's plane
            lib.lsb_sell_tmpls_free(T), lib.lsb_sell_vc_free(V), lib.lsb_sell_free(H)
    # a 2-D operator has no plane-reaching far slots
    A = la.lsbench_matrix_synth("lap2d:nx=411,ny=203")
    H = lib.lsb_csr_sellize16(A.ptr, 0)
    V = lib.lsb_sell16_value_slots(H)
    T = lib.lsb_sell16_templates(H, V)
    assert T and not lib.lsb_sell_tmpl_columns(T, 16, 8)
    lib.lsb_sell_tmpls_free(T), lib.lsb_sell_vc_free(V), lib.lsb_sell_free(H)


def test_line_padding_of_a_2d_grid():
    """lsb_csr_pad_lines: a constant-coefficient 2-D grid whose lines are not whole slices, re-numbered so
    that every line starts at a multiple of 128 rows.  The real block of the padded operator IS the
    operator (entry for entry under the row map), the pad rows couple only among themselves (the
    same stencil on a strip: block-diagonal, SPD), so with b = 0 there the solve is the same solve;
    and the padded copy has what the unpadded one lacks -- +-nx as whole-slice offsets: shaped
    templates with ONE far slot per side whose planes are grid lines, i.e. a z-column plan along y.
    Refused: general values, 3-D grids, lines that are whole slices already, padding above 1/16."""
    import ctypes as C
    import scipy.sparse as sp
    lib = la._lib.load()
    A = la.lsbench_matrix_synth("lap2d:nx=1000,ny=64")
    nx, nxp = C.c_uint(0), C.c_uint(0)
    mp = C.POINTER(C.c_int)()
    P = lib.lsb_csr_pad_lines(A.ptr, 128, C.byref(nx), C.byref(nxp), C.byref(mp))
    assert P and (nx.value, nxp.value) == (1000, 1024)
    p = P.contents
    n, npad = A.nrows, p.nrows
    assert npad == 64 * 1024
    offs = np.ctypeslib.as_array(p.offs, (npad + 1,)).copy()
    cols = np.ctypeslib.as_array(p.cols, (offs[-1],)).copy()
    vals = np.ctypeslib.as_array(p.vals, (offs[-1],)).copy()
    m = np.ctypeslib.as_array(mp, (npad,)).copy()
    M = sp.csr_matrix((vals, cols, offs), shape=(npad, npad))
    S = sp.csr_matrix((A.vals, A.cols, A.offs), shape=(n, n))
    real = np.flatnonzero(m >= 0)
    assert np.array_equal(m[real], np.arange(n))                   # every row once, in order
    assert (M[real][:, real] != S).nnz == 0                        # the real block is the operator
    pad = np.flatnonzero(m < 0)
    assert M[real][:, pad].nnz == 0 and M[pad][:, real].nnz == 0   # block-diagonal
    Pb = M[pad][:, pad]
    assert (Pb != Pb.T).nnz == 0 and np.all(Pb.diagonal() == 4.0) and Pb.nnz == 5 * len(pad) - 2 * 24 - 2 * 64
    assert np.linalg.eigvalsh(Pb.toarray()[:480, :480]).min() > 0
    H = lib.lsb_csr_sellize16(P, 0)
    V = lib.lsb_sell16_value_slots(H)
    T = lib.lsb_sell16_templates(H, V)
    assert T and T.contents.nfar == 1
    Cp = lib.lsb_sell_tmpl_columns(T, 8, 4)                        # a line = 8 slices
    assert Cp and Cp.contents.in_cols * 10 >= T.contents.nslice * 9 and Cp.contents.centre0 == 1
    why = C.create_string_buffer(256)
    assert lib.lsb_tmpl_cols_check(T, Cp, why, 256) == 0, why.value
    lib.lsb_tmpl_cols_free(Cp), lib.lsb_sell_tmpls_free(T), lib.lsb_sell_vc_free(V), lib.lsb_sell_free(H)
    la._lib.libc_free(mp)
    lib.lsb_csr_free(P)
    for spec in ("lap2d:nx=1000,ny=64,coef=1", "lap3d:nx=40,ny=36,nz=30", "lap2d:nx=1024,ny=64", "lap2d:nx=411,ny=203"):
        B = la.lsbench_matrix_synth(spec)
        assert not lib.lsb_csr_pad_lines(B.ptr, 128, None, None, None), spec


def test_bench_quotes_pmc_traffic_only_for_what_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from profiles/pmc_traffic.json -- a LIST of profiled flavours
    per workload -- and is quoted only where kernel name, hash of the kernel sources, kept / all value
    slots AND the flavour the timing pass picked (spmv_flags, xcd_period_slices) are those of the run;
    anything else reads null with the reason."""
    import importlib.util, json, os, sys
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    saved = sys.argv
    sys.argv = ["bench.py"]
    try:
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
    finally:
        sys.argv = saved
    sha = bench.kernels_sha16()
    (tmp_path / "profiles").mkdir()
    entries = {"lap3d": [dict(bytes=1833e6, kernel="k_spmv_tmpl", kernels_sha16=sha, value_slots=[3, 9], spmv_flags=198,
                              xcd_period_slices=0, source="a.csv"),
                         dict(bytes=1389e6, kernel="k_spmv_tmpl", kernels_sha16=sha, value_slots=[3, 9], spmv_flags=198,
                              xcd_period_slices=1250, source="b.csv")],
               "old": dict(bytes=1.0, kernel="k_spmv_tmpl", kernels_sha16="0" * 16, value_slots=[3, 9], spmv_flags=70,
                           xcd_period_slices=0)}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps(entries))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernels_sha16", lambda: sha)
    assert bench.pmc_traffic("lap3d", "k_spmv_tmpl", (3, 9), 198, 1250) == (1389e6, "b.csv")
    assert bench.pmc_traffic("lap3d", "k_spmv_tmpl", (3, 9), 198, 0) == (1833e6, "a.csv")
    t, why = bench.pmc_traffic("lap3d", "k_spmv_tmpl", (3, 9), 70, 0)
    assert t is None and "other flavour" in why and "(198, 1250)" in why
    t, why = bench.pmc_traffic("lap3d", "k_spmv_sell16", (3, 9), 198, 0)
    assert t is None and "k_spmv_sell16" in why
    t, why = bench.pmc_traffic("lap3d", "k_spmv_tmpl", (4, 9), 198, 0)
    assert t is None and "value slots" in why
    t, why = bench.pmc_traffic("old", "k_spmv_tmpl", (3, 9), 70, 0)          # (a dict of earlier rounds: one entry)
    assert t is None and "kernel sources" in why
    assert bench.pmc_traffic("nothing", "k", (0, 0), 0, 0)[0] is None and bench.pmc_traffic(None, "k", (0, 0))[0] is None
    # GMRES(m) bytes: the first step of a cycle reads one basis column per Gram-Schmidt pass, the 30th thirty
    n, sp = 1000, 5000
    one = bench.gmres_bytes(n, sp, 30, 1)
    assert one == 24 * n + 8 * n * 4 + 32 * n + sp + 2 * 8 * n * 2 + 2 * 8 * n * 3
    assert bench.gmres_bytes(n, sp, 30, 31) - bench.gmres_bytes(n, sp, 30, 30) == one + 16 * n + sp + 8 * n * (1 - 30) + 8 * n * 30 - 8 * n
