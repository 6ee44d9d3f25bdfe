"""Shared fixtures.  `-m "not gpu"` = oracle, host logic, C-ABI surface (runs
in the build container); `-m gpu` = parity of the HIP path against the oracle
and the golden vectors, all through the C-ABI (runs on an MI355X)."""
import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")
TOY = ["A0_02x02", "A1_02x02", "I1_05x05"]
SPD = ["xn3b_A_18", "xn3b_A_15", "xn3b_A_12", "xn3b_A_10",
       "tj7a_A_18", "tj7a_A_15", "tj7a_A_12"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build what is missing (library, oracle); never the reference on the GPU box."""
    from lsbench_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    from oracle import oracle
    if not os.path.exists(os.path.join(ROOT, "oracle", "liblsb_oracle.so")):
        oracle.build()


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLD, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def matrix_path(tmp_path_factory):
    """Path of a reference test matrix (the big ones are stored gzipped)."""
    cache = tmp_path_factory.mktemp("matrices")

    def get(name):
        p = os.path.join(GOLD, "matrices", name + ".txt")
        if os.path.exists(p):
            return p
        out = os.path.join(str(cache), name + ".txt")
        if not os.path.exists(out):
            with gzip.open(p + ".gz", "rb") as fi, open(out, "wb") as fo:
                fo.write(fi.read())
        return out
    return get


@pytest.fixture(scope="session")
def golden_x():
    import numpy as np

    def get(name):
        return np.fromfile(os.path.join(GOLD, "x", name + ".x.f64"), dtype="<f8")
    return get


@pytest.fixture(scope="session")
def hip():
    """Initialised backend.  No GPU => the test FAILS (it is marked gpu)."""
    import torch  # noqa: F401  (first, so the library binds torch's HIP runtime)
    import lsbench_amd as la
    rc = la.hip_cdna4_init()
    assert rc == 0 or la._lib.load().lsb_hip_stream(), \
        "hip_cdna4_init failed: no GPU visible -- gpu tests must run on an MI355X"
    yield la
