"""Compile-time guard for the hot kernels (no GPU needed: hipcc cross-compiles
gfx950): no scratch spills, <= 64 VGPRs (8 wavefronts/SIMD), 16 KiB of LDS per
workgroup for the adaptive SpMV.  An innocent edit of a rare branch once pushed
k_spmv_adaptive into scratch and cost 70 % of its bandwidth; this catches it at
build time."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "lsbench_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="hipcc not installed")
def test_hot_kernels_have_no_spills_and_full_occupancy(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c",
                        os.path.join(CSRC, "hip_kernels.hip"), "-o", str(tmp_path / "k.o"),
                        "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    info, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            info[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            info[name][m.group(1).strip()] = int(m.group(2))
    hot = {k: v for k, v in info.items()
           if re.search(r"k_spmv_adaptive|k_pcg_update_xr|k_pcg_update_p|k_pcg_init", k)}
    assert len([k for k in hot if "k_spmv_adaptive" in k]) == 8  # four flavours x {fp64, fp32 values}
    for k, v in hot.items():
        assert v["ScratchSize"] == 0, (k, v)
        assert v["VGPRs"] <= 64, (k, v)
        assert v["Occupancy"] == 8, (k, v)
        if "k_spmv_adaptive" in k:
            assert v["LDS Size"] <= 16 * 1024 + 256, (k, v)   # products + reductions + the folded all-reduce tail
    sell = {k: v for k, v in info.items() if "k_spmv_sell" in k}
    # {32-bit, 16-bit columns} x {plain, nontemporal} x {fp64, fp32} + the four 16-bit ones with the
    # Chebyshev step in the epilogue
    assert len(sell) == 12
    for k, v in sell.items():                                     # no LDS staging, >= 6 workgroups per CU
        assert v["ScratchSize"] == 0 and v["VGPRs"] <= 80 and v["Occupancy"] >= 6, (k, v)
        assert v["LDS Size"] <= 192, (k, v)     # reductions + the folded all-reduce tail; no staging
    col = {k: v for k, v in info.items() if "k_spmv_tmpl_col" in k}
    # the z-column walk: 1, 2 far slots per side x {no dot, the centre pair as the dot's operand, a loaded one}
    assert len(col) == 6
    for k, v in col.items():                                      # registers only: three planes of centre pairs
        assert v["ScratchSize"] == 0 and v["VGPRs"] <= 96 and v["Occupancy"] >= 5 and v["LDS Size"] <= 192, (k, v)
    tm = {k: v for k, v in info.items() if "k_spmv_tmpl" in k and k not in col}
    # 0, 1, 2 far slots per side x {plain, Chebyshev epilogue, plain with the deferred store}
    assert len(tm) == 9
    for k, v in tm.items():                                       # straight-line gathers, >= 6 workgroups per CU
        assert v["ScratchSize"] == 0 and v["VGPRs"] <= 80 and v["Occupancy"] >= 6, (k, v)
        # the deferred-store form parks 16 B per lane in LDS (4 KB); nothing else is staged
        assert v["LDS Size"] <= (4096 + 192 if v["LDS Size"] > 192 else 192), (k, v)
    assert sum(v["LDS Size"] > 192 for v in tm.values()) == 3
    for k, v in info.items():
        assert v.get("ScratchSize", 0) == 0, (k, v)
