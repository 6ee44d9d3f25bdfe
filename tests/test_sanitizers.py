"""the reference has no sanitizer build; here the host C side
that builds the kernels' layouts runs under ASan + UBSan (CPU build only -- GPU
AddressSanitizer is not available on the pool)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "lsbench_amd", "csrc")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not installed")
def test_host_layout_builders_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_host")
    r = subprocess.run(["gcc", "-g", "-O1", "-std=gnu11", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-fopenmp",
                        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include",
                        "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "asan_host.c"),
                        os.path.join(CSRC, "lsb_operator.c"), os.path.join(CSRC, "lsb_synth.c"),
                        "-o", exe, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", OMP_NUM_THREADS="4")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    # seven operators + the two grids of the padding / z-column block
    assert r.stdout.count("ok ") == 9 and "ERROR" not in r.stderr and "runtime error" not in r.stderr
