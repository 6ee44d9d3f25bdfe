#!/usr/bin/env python3
"""Generate tests/golden/ -- run in the BUILD container only (needs /root/reference).

ORACLE tooling, test infrastructure only.  Writes DATA, never reference source:

  tests/golden/matrices/*.txt[.gz]   the reference's own test matrices
                                     (/root/reference/tests/*.txt), copied
                                     verbatim as data files (the big ones gzipped)
  tests/golden/loader_cases/*.txt    small hand-made inputs exercising the loader
                                     rules of src/lsbench-csr.c:29-92
  tests/golden/ref_print/*.print     what the REFERENCE's own loader+printer
                                     (oracle/_ref/liblsbench_ref.so, compiled from
                                     /root/reference/src by oracle/Makefile) prints
                                     for each of those inputs (big ones: md5 only)
  tests/golden/x/*.x.f64             golden solutions x = S^-1 b, raw little-endian
                                     float64, S = triu(A)+triu(A,1)^T
                                     (src/cholmod-impl.h:5-21), b_i = i
                                     (src/lsbench.c:159-160); dense LAPACK Cholesky
                                     cross-checked against SuperLU
  tests/golden/golden.json           sizes, norms, spreads, md5s, PCG iteration counts

CHOLMOD itself cannot be run (not in /root/reference, no network), so the golden
x are pinned by the mathematical definition, not by a reference output.
"""
import ctypes
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import scipy.linalg as sl
import scipy.sparse as sp
import scipy.sparse.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF_TESTS = "/root/reference/tests"
GOLD = os.path.join(ROOT, "tests", "golden")
REFLIB = os.path.join(ROOT, "oracle", "_ref", "liblsbench_ref.so")

TOY = ["A0_02x02", "A1_02x02", "I1_05x05"]
SPD = ["xn3b_A_18", "xn3b_A_15", "xn3b_A_12", "xn3b_A_10",
       "tj7a_A_18", "tj7a_A_15", "tj7a_A_12"]

LOADER_CASES = {
    # unsorted + duplicate (1,2): duplicates are summed (src/lsbench-csr.c:57-63)
    "dup_unsorted_b1": "6 1\n2 2 4.0\n1 2 1.0\n1 1 3.0\n2 1 1.0\n1 2 -0.25\n2 2 0.5\n",
    # base 0, unsorted rows
    "unsorted_b0": "5 0\n2 2 1.5\n0 0 2.0\n1 1 3.0\n0 2 -1.0\n2 0 -1.0\n",
    # row id 2 absent: rows are renumbered densely, columns kept (:66-70)
    "missing_row_b1": "3 1\n1 1 2.0\n3 3 5.0\n3 1 -1.0\n",
    # scientific notation / negative values through %lf
    "sci_values_b1": "3 1\n1 1 1e-3\n2 2 -2.5E+2\n2 1 3\n",
}


def ref_print(path):
    """stdout of the reference's lsbench_matrix_print(lsbench_matrix_read(path))."""
    code = (
        "import ctypes,sys;L=ctypes.CDLL(%r);L.lsbench_matrix_read.restype=ctypes.c_void_p;"
        "L.lsbench_matrix_read.argtypes=[ctypes.c_char_p];L.lsbench_matrix_print.argtypes=[ctypes.c_void_p];"
        "A=L.lsbench_matrix_read(%r.encode());L.lsbench_matrix_print(A);"
        "ctypes.CDLL(None).fflush(None)" % (REFLIB, path))
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True)
    return out.stdout


def md5(b):
    return hashlib.md5(b).hexdigest()


def main():
    if not os.path.exists(REFLIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    for d in ("matrices", "loader_cases", "ref_print", "x"):
        os.makedirs(os.path.join(GOLD, d), exist_ok=True)
    meta = {"matrices": {}, "loader_cases": {}}

    # --- loader cases -----------------------------------------------------
    for name, text in LOADER_CASES.items():
        p = os.path.join(GOLD, "loader_cases", name + ".txt")
        with open(p, "w") as f:
            f.write(text)
        out = ref_print(p)
        with open(os.path.join(GOLD, "ref_print", name + ".print"), "wb") as f:
            f.write(out)
        meta["loader_cases"][name] = {"print_md5": md5(out)}

    # --- reference matrices ------------------------------------------------
    for name in TOY + SPD:
        src = os.path.join(REF_TESTS, name + ".txt")
        big = name in SPD
        if big:
            with open(src, "rb") as fi, gzip.GzipFile(
                    os.path.join(GOLD, "matrices", name + ".txt.gz"), "wb",
                    compresslevel=9, mtime=0) as fo:
                shutil.copyfileobj(fi, fo)
        else:
            shutil.copyfile(src, os.path.join(GOLD, "matrices", name + ".txt"))
        out = ref_print(src)
        if not big:
            with open(os.path.join(GOLD, "ref_print", name + ".print"), "wb") as f:
                f.write(out)
        A = O.matrix_read(src)
        S = O.operator_upper(A)
        S2 = O.operator_upper_numpy(A)
        assert np.array_equal(S.cols, S2.indices) and np.array_equal(S.vals, S2.data)
        n = A.nrows
        b = O.rhs(n)
        entry = {"n": n, "base": A.base, "nnz": A.nnz, "nnz_S": S.nnz,
                 "print_md5": md5(out), "file_md5": md5(open(src, "rb").read())}
        Sd = S2.toarray()
        entry["asym_A"] = float(abs(A.to_scipy() - A.to_scipy().T).max())
        if name in SPD:
            cf = sl.cho_factor(Sd, lower=True)
            x = sl.cho_solve(cf, b)
            # one step of iterative refinement in extended precision
            r = (b.astype(np.longdouble) - (Sd.astype(np.longdouble) @ x.astype(np.longdouble))).astype(np.float64)
            x = x + sl.cho_solve(cf, r)
            method = "dense LAPACK Cholesky (dpotrf/dpotrs) + 1 refinement step"
        else:
            x = np.linalg.solve(Sd, b)
            method = "dense LAPACK LU (dgesv)"
        lu = sla.splu(sp.csc_matrix(S2), permc_spec="MMD_AT_PLUS_A",
                      options=dict(SymmetricMode=True))
        x_lu = lu.solve(b)
        nx = float(np.linalg.norm(x))
        entry["method"] = method
        entry["x_norm2"] = nx
        entry["spread_vs_superlu"] = float(np.linalg.norm(x - x_lu) / nx) if nx else 0.0
        entry["relres"] = float(np.linalg.norm(b - S2 @ x) / max(np.linalg.norm(b), 1e-300))
        # what solving the RAW file matrix instead of S would give (the parity trap)
        x_raw = sla.spsolve(sp.csc_matrix(A.to_scipy()[:, :n]), b)
        entry["raw_vs_S"] = float(np.linalg.norm(x_raw - x) / nx) if nx else 0.0
        xs, it, rel, st = O.pcg_jacobi(S.offs, S.cols, S.vals, b, 1e-12, 20000)
        entry["pcg_tol1e-12"] = {"iters": it, "relres": rel, "status": st,
                                 "err_vs_golden": float(np.linalg.norm(xs - x) / nx) if nx else 0.0}
        x.astype("<f8").tofile(os.path.join(GOLD, "x", name + ".x.f64"))
        meta["matrices"][name] = entry
        print(name, json.dumps(entry))

    with open(os.path.join(GOLD, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
