"""The wiring a maintainer adds to thilinarmtb/lsbench, proven on a scratch copy of
the reference (build container only: /root/reference does not exist on the GPU
box).  integration/hip.patch + integration/src/hip_cdna4.c + integration/libs/
hip.cmake are applied to a COPY of the reference tree in a temp dir (nothing of
the reference is copied into this repository), the tree is compiled with plain
gcc/g++ with every other backend off -- the recipe of oracle/Makefile `ref`,
not the reference's CMake -- once with -DLSBENCH_HIP linked against this
repository's liblsbench_hip.so and once without (the disabled-backend stubs),
and `driver --solver hip` is run on a reference matrix.  The CMake leg is run as
well: the reference's own CMakeLists.txt (as patched) + integration/libs/hip.cmake are
configured with -DENABLE_HIP=ON / OFF (every downloaded backend off;
--compile-no-warning-as-error because the reference's -Werror trips on its own
src/lsbench.c:70 with this gcc), built, and the built driver is
run -- with integration/hip-flags.patch on top, which gives the reference's own
lsbench_init the backend's flags (--tol, --maxit, --ngpus, ...) and its
lsbench_matrix_read the `synth:` prefix, so that BASELINE configs 3-5 are reachable
through the real driver.
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


def _patched_tree(tmp_path, flags=False):
    tree = tmp_path / "lsbench"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns("tests"))
    for name in ["hip.patch"] + (["hip-flags.patch"] if flags else []):
        r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(INTEG, name)],
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


def _cmake(tree, bdir, hip):
    """The reference's own build: CMakeLists.txt:5-10,22-25,32-55 as patched + libs/hip.cmake."""
    cfg = subprocess.run(["cmake", "-S", str(tree), "-B", str(bdir), "-DENABLE_CHOLMOD=OFF",
                          "-DENABLE_HIP=" + ("ON" if hip else "OFF"), "-DLSBENCH_HIP_ROOT=" + ROOT,
                          "--compile-no-warning-as-error"], capture_output=True, text=True)
    assert cfg.returncode == 0, cfg.stdout[-2000:] + cfg.stderr[-2000:]
    bld = subprocess.run(["cmake", "--build", str(bdir)], capture_output=True, text=True)
    assert bld.returncode == 0, bld.stdout[-2000:] + bld.stderr[-3000:]
    return bdir / "liblsbench.so", bdir / "driver"


@pytest.mark.skipif(shutil.which("cmake") is None, reason="no cmake")
@pytest.mark.parametrize("flags", [False, True])
def test_cmake_leg_configures_builds_and_runs(tmp_path, flags):
    """cmake -DENABLE_HIP=ON -DLSBENCH_HIP_ROOT=<this repository> on the patched reference:
    libs/hip.cmake finds and links liblsbench_hip.so, the built driver runs `--solver hip`;
    -DENABLE_HIP=OFF builds the stubs.  With hip-flags.patch the reference's CLI takes the
    backend's flags and `--matrix synth:SPEC`."""
    tree = _patched_tree(tmp_path, flags=flags)
    matrix = os.path.join(REF, "tests", "I1_05x05.txt")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    lib1, drv1 = _cmake(tree, tmp_path / "build_on", hip=True)
    ldd = subprocess.run(["ldd", str(lib1)], capture_output=True, text=True, env=env).stdout
    assert os.path.join(CSRC, "liblsbench_hip.so") in ldd           # found through LSBENCH_HIP_ROOT, rpath set
    nm = subprocess.run(["nm", "-D", str(lib1)], capture_output=True, text=True).stdout
    assert " U hip_cdna4_bench" in nm and " T lsbench_bench" in nm and " T hip_cdna4_init" not in nm
    import lsbench_amd as la
    gpu = la._lib.load().lsb_hip_device_count() > 0
    r = subprocess.run([str(drv1), "--solver", "hip", "--matrix", matrix, "--trials=2"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert ("===matrix,n,nnz,trials,solver,ordering,elapsed===" in r.stdout) == gpu
    if flags:
        assert " U hip_cdna4_set_option" in nm and " U hip_cdna4_matrix_synth" in nm
        # BASELINE config 3's generator through the REAL driver, the backend's flags on its command line
        r = subprocess.run([str(drv1), "--solver", "hip", "--matrix", "synth:lap2d:nx=40,ny=30",
                            "--operator", "raw", "--tol", "1e-9", "--maxit", "500", "--ngpus", "1",
                            "--krylov", "cg", "--trials=1"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        if gpu:
            rec = r.stdout.splitlines()
            k = rec.index("===hip_cdna4:iterations,relres,status,tol,solves_per_sec,nshards===")
            f = rec[k + 1].split(",")
            assert int(f[2]) == 1 and float(f[3]) == 1e-9 and "synth:lap2d:nx=40,ny=30,1200,5860,1,6," in r.stdout
        # a value the option does not take, a spec the generator does not know: exit 1 with a message
        r = subprocess.run([str(drv1), "--solver", "hip", "--matrix", matrix, "--krylov", "qmr"],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 1 and "not a value of option `krylov'" in r.stderr
        r = subprocess.run([str(drv1), "--solver", "hip", "--matrix", "synth:hilbert:n=4"],
                           capture_output=True, text=True, env=env)
        assert r.returncode == 1 and "Unable to generate" in r.stderr
    # ENABLE_HIP=OFF: the stub file is what defines the symbols; `--solver hip` is the silent no-op
    lib0, drv0 = _cmake(tree, tmp_path / "build_off", hip=False)
    nm0 = subprocess.run(["nm", "-D", str(lib0)], capture_output=True, text=True).stdout
    assert " T hip_cdna4_bench" in nm0 and "lsbench_hip" not in subprocess.run(
        ["ldd", str(lib0)], capture_output=True, text=True).stdout
    r = subprocess.run([str(drv0), "--solver", "hip", "--matrix", matrix], capture_output=True, text=True)
    assert r.returncode == 0 and "===matrix" not in r.stdout
    if flags:
        r = subprocess.run([str(drv0), "--solver", "hip", "--matrix", "synth:lap2d:nx=9,ny=4"],
                           capture_output=True, text=True)
        assert r.returncode == 1 and "ENABLE_HIP off" in r.stderr
