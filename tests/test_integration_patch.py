"""The wiring a maintainer adds to thilinarmtb/lsbench, proven on a scratch copy of
the reference (build container only: /root/reference does not exist on the GPU
box).  integration/hip.patch + integration/src/hip_cdna4.c + integration/libs/
hip.cmake are applied to a COPY of the reference tree in a temp dir (nothing of
the reference is copied into this repository), the tree is compiled with plain
gcc/g++ with every other backend off -- the recipe of oracle/Makefile `ref`,
not the reference's CMake -- once with -DLSBENCH_HIP linked against this
repository's liblsbench_hip.so and once without (the disabled-backend stubs),
and `driver --solver hip` is run on a reference matrix.
Reference wiring points: CMakeLists.txt:5-10,22-25,32-55; src/lsbench.c:15-35,
73-74,143-147,162-184,190-194; stub convention src/cholmod.c:74-81."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
INTEG = os.path.join(ROOT, "integration")
CSRC = os.path.join(ROOT, "lsbench_amd", "csrc")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")),
                                reason="the reference tree is only present in the build container")

C_FILES = ["lsbench.c", "lsbench-csr.c", "cusparse.c", "hypre.c", "amgx.c", "cholmod.c", "hip_cdna4.c"]
CXX_FILES = ["paralmond.cpp", "ginkgo.cpp"]


def _patched_tree(tmp_path):
    tree = tmp_path / "lsbench"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns("tests"))
    r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(INTEG, "hip.patch")],
                       cwd=tree, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    shutil.copy(os.path.join(INTEG, "src", "hip_cdna4.c"), tree / "src" / "hip_cdna4.c")
    shutil.copy(os.path.join(INTEG, "libs", "hip.cmake"), tree / "libs" / "hip.cmake")
    return tree


def _build(tree, out, with_hip):
    out.mkdir()
    defs = ["-DLSBENCH_HIP"] if with_hip else []
    objs = []
    for f in C_FILES:
        o = out / (f + ".o")
        subprocess.run(["gcc", "-std=gnu11", "-O1", "-w", "-fPIC", "-I", str(tree / "src")] + defs +
                       ["-c", str(tree / "src" / f), "-o", str(o)], check=True)
        objs.append(str(o))
    for f in CXX_FILES:
        o = out / (f + ".o")
        subprocess.run(["g++", "-std=gnu++17", "-O1", "-w", "-fPIC", "-I", str(tree / "src")] + defs +
                       ["-c", str(tree / "src" / f), "-o", str(o)], check=True)
        objs.append(str(o))
    link = ["-L", CSRC, "-llsbench_hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"] if with_hip else []
    lib = out / "liblsbench.so"
    subprocess.run(["g++", "-shared", "-o", str(lib)] + objs + link, check=True)
    drv = out / "driver"
    subprocess.run(["gcc", "-O1", "-w", "-I", str(tree / "src"), str(tree / "bin" / "driver.c"), "-o", str(drv),
                    "-L", str(out), "-llsbench", "-Wl,-rpath," + str(out)] + link, check=True)
    return lib, drv


def test_patch_applies_and_touches_the_seven_wiring_points(tmp_path):
    tree = _patched_tree(tmp_path)
    src = (tree / "src" / "lsbench.c").read_text()
    assert "LSBENCH_SOLVER_HIP = 6" in (tree / "src" / "lsbench.h").read_text()           # (1)
    assert "int hip_cdna4_bench(" in (tree / "src" / "lsbench-impl.h").read_text()        # (2)
    assert 'strcmp(up, "HIP") == 0' in src                                                # (3)
    assert "ginkgo, hip" in src                                                           # (4)
    assert "hip_cdna4_init();" in src and "case LSBENCH_SOLVER_HIP:" in src               # (5) (6)
    assert "hip_cdna4_finalize();" in src                                                 # (7)
    cm = (tree / "CMakeLists.txt").read_text()
    assert "option(ENABLE_HIP" in cm and "src/hip_cdna4.c" in cm and "include(libs/hip.cmake)" in cm
    assert "-DLSBENCH_HIP" in cm


def test_patched_reference_builds_links_and_runs(tmp_path):
    tree = _patched_tree(tmp_path)
    matrix = os.path.join(REF, "tests", "I1_05x05.txt")
    # ENABLE_HIP=OFF: the stubs -> `--solver hip` is the reference's silent no-op
    lib0, drv0 = _build(tree, tmp_path / "off", with_hip=False)
    nm = subprocess.run(["nm", "-D", str(lib0)], capture_output=True, text=True).stdout
    assert " T hip_cdna4_bench" in nm                                  # defined by the stub file
    r = subprocess.run([str(drv0), "--solver", "hip", "--matrix", matrix], capture_output=True, text=True)
    assert r.returncode == 0 and "===matrix" not in r.stdout
    # ENABLE_HIP=ON: the three symbols come from this repository's backend library
    lib1, drv1 = _build(tree, tmp_path / "on", with_hip=True)
    nm = subprocess.run(["nm", "-D", str(lib1)], capture_output=True, text=True).stdout
    assert " U hip_cdna4_bench" in nm and " T lsbench_bench" in nm
    ldd = subprocess.run(["ldd", str(lib1)], capture_output=True, text=True).stdout
    assert "liblsbench_hip.so" in ldd
    # ... and nothing of the reference's own API is defined twice
    back = subprocess.run(["nm", "-D", "--defined-only", os.path.join(CSRC, "liblsbench_hip.so")],
                          capture_output=True, text=True).stdout
    for sym in ("lsbench_init", "lsbench_bench", "lsbench_finalize", "lsbench_matrix_read",
                "lsbench_matrix_print", "lsbench_matrix_free", "lsbench_get_matrix_name"):
        assert (" T %s\n" % sym) not in back, sym
    env = dict(os.environ, LD_LIBRARY_PATH=CSRC + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(drv1), "--solver", "hip", "--matrix", matrix, "--trials=2"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    import lsbench_amd as la
    if la._lib.load().lsb_hip_device_count() > 0:      # a GPU box with the reference mounted: it solves
        assert "===matrix,n,nnz,trials,solver,ordering,elapsed===" in r.stdout
        assert "5,5,2,6," in r.stdout
    else:                                              # no GPU: quiet return, like any backend that
        assert "===matrix" not in r.stdout             # could not initialise (src/cusparse.c:166-167)
    # the other solvers of the patched tree are untouched
    r = subprocess.run([str(drv1), "--solver", "cholmod", "--matrix", matrix], capture_output=True, text=True,
                       env=env)
    assert r.returncode == 0
