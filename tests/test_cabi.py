"""The C-ABI surface: the shared library loads, exports every symbol that
include/*.h declares, lays its structs out like the reference, and refuses
loudly (return code 1, no CPU path) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import lsbench_amd as la
from conftest import ROOT
from lsbench_amd import _lib


def _declared_functions():
    names = set()
    for h in ("lsbench.h", "lsbench_hip.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
        for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", text, flags=re.S):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_functions()
    assert {"hip_cdna4_init", "hip_cdna4_finalize", "hip_cdna4_bench",
            "lsbench_matrix_read", "lsb_hip_spmv_csr_f64"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), "declared in include/ but not exported: " + name
    # and the binding covers exactly the header
    assert set(_lib.SIGNATURES) == declared


def test_struct_layouts_match_reference():
    # src/lsbench-impl.h:14-26 on LP64
    assert C.sizeof(_lib.CsrStruct) == 32 and _lib.CsrStruct.offs.offset == 8
    assert _lib.CsrStruct.cols.offset == 16 and _lib.CsrStruct.vals.offset == 24
    assert C.sizeof(_lib.LsbenchStruct) == 32 and _lib.LsbenchStruct.solver.offset == 8
    assert _lib.LsbenchStruct.trials.offset == 24
    # enum values (src/lsbench.h:8-29) + the one addition
    text = open(os.path.join(ROOT, "include", "lsbench.h")).read()
    for name, val in [("CUSOLVER", 0), ("HYPRE", 1), ("AMGX", 2), ("CHOLMOD", 3),
                      ("PARALMOND", 4), ("GINKGO", 5), ("HIP", 6)]:
        assert re.search(r"LSBENCH_SOLVER_%s = %d\b" % (name, val), text)


def test_opts_defaults():
    o = la.default_opts()
    assert (o.tol, o.maxit, o.op_mode, o.precond, o.nvirt) == (1e-12, 20000, 0, 0, 1)


@pytest.mark.skipif(_lib.load().lsb_hip_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_refusal_not_fallback():
    lib = _lib.load()
    assert lib.hip_cdna4_init() == 1      # quiet, like a disabled backend
    assert lib.hip_cdna4_finalize() == 1
    A = la.lsbench_matrix_synth("lap2d:nx=4,ny=4")
    with pytest.raises(la.LsbenchHipError, match="no CPU path"):
        la.hip_cdna4_bench(A)
    with pytest.raises(la.LsbenchHipError):
        la.Solver(A)
    buf = np.zeros(4)
    for rc in (lib.lsb_hip_dot_f64(4, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data,
                                   buf.ctypes.data, None),
               lib.lsb_hip_axpy_f64(4, buf.ctypes.data, buf.ctypes.data, buf.ctypes.data, None),
               lib.lsb_hip_sync()):
        assert rc == 1


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under lsbench_amd/, include/
    may import, link or name it."""
    for base in ("lsbench_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".c", ".h", ".hip", "Makefile")):
                    text = open(os.path.join(d, f), errors="ignore").read()
                    assert "liblsb_oracle" not in text and "import oracle" not in text \
                        and "from oracle" not in text, os.path.join(d, f)
    import subprocess
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip64" in out and "rccl" in out
