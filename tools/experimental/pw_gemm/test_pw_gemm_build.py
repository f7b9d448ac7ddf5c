"""Build-time invariant of tools/experimental/pw_gemm/pw_gemm.hip (no GPU; run by path: `pytest tools/experimental/pw_gemm`): none of its kernels may use scratch.  The MFMA waves read their
operand fragments by inline-asm ds_read_b128 whose results become valid only at a later s_waitcnt; a register spill (or any
compiler-inserted copy) between the two would save a stale value.  Zero scratch is the part of that contract a test can
hold."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
PW = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "audio_visual_deepfake_detection_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_pw_gemm_kernels_use_no_scratch(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = tmp_path / "pw_gemm.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on", "-S", "--cuda-device-only",
                    "-I", CSRC, "-I", PW, "-I", os.path.join(ROOT, "include"), os.path.join(PW, "pw_gemm.hip"), "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    asm = out.read_text()
    kernels = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", asm, re.S)
    assert len(kernels) >= 10, "expected one kernel per tile configuration and residual switch"
    for name, body in kernels:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body).group(1))
        assert scratch == 0, f"{name}: {scratch} bytes of scratch"
        lds = int(re.search(r"\.amdhsa_group_segment_fixed_size\s+(\d+)", body).group(1))
        assert lds <= 160 * 1024, f"{name}: {lds} bytes of LDS"
