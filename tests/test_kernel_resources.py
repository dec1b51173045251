"""Compile-time properties the measured performance of the timed kernels rests on (no GPU needed: hipcc
cross-compiles for gfx950).  Round 2 found the f32 kernel at 256 VGPRs + 84 AGPRs, one wave per SIMD, with a
v_accvgpr_read per element in its distance chain, because its row registers were defined under one exec mask and
consumed under another; this test keeps that from coming back unnoticed."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_lean_kernels_fit_two_waves_per_simd_without_agprs_or_scratch(tmp_path):
    src = os.path.join(ROOT, "hnsw_rs_amd", "csrc", "search_lean.hip")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
           "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", str(tmp_path / "lean.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    timed = {k: v for k, v in kernels.items() if "hx_lean_f32_kernel" in k or "hx_lean_q8_kernel" in k}
    # f32 100d and quant8: one register, head + tail, interleaved two (A/B), interleaved four (ef 129..256); f32 128d
    # (cooperative gather): one register and head + tail at two stage depths each, interleaved four
    assert len(timed) == 23, sorted(kernels)  # (+ five to eight interleaved registers for ef 257..512: f32 100d, quant8; six / eight for f32 128d)
    # the two-wave form (pair_kernel.inc, opt-in): one register and head + tail; a workgroup is two waves, four
    # workgroups per CU by LDS, so it must stay within 256 registers without scratch
    pair = {k: v for k, v in kernels.items() if "hx_pair_f32_kernel" in k}
    assert len(pair) == 2, sorted(kernels)
    for name, r in pair.items():
        assert r.get("ScratchSize", 0) == 0 and r.get("VGPRs Spill", 0) == 0 and r.get("AGPRs", 0) == 0, (name, r)
        assert r.get("VGPRs", 999) <= 256, (name, r)
    for name, r in timed.items():
        assert r.get("ScratchSize", 0) == 0 and r.get("VGPRs Spill", 0) == 0, (name, r)
        if "ILi128E" in name or ("hx_lean_f32_kernel" in name and re.search(r"LstILi[5678]E", name)):
            # d = 128 holds the query's 128 values in registers beside the gather's stage ring: it sits AT the
            # 256-register line (a couple of values parked in AGPRs), which costs nothing at the metric's one wave
            # per SIMD; what must not come back is scratch or a ring that lives in AGPRs
            # (the six- to eight-register lists carry the second visited level's state too: 7 / 18 / 20 values in
            # AGPRs at one wave per SIMD, which is what a batch of 1024 runs at)
            wide = re.search(r"LstILi[678]E", name)
            assert r.get("AGPRs", 0) <= ((40 if "ILi128E" in name else 24) if wide else 16), (name, r)
            continue
        assert r.get("AGPRs", 0) == 0, (name, r)
        assert r.get("Occupancy", 0) >= 2, (name, r)
        assert r.get("VGPRs", 999) <= 248, (name, r)  # (the four-register list: 244)
