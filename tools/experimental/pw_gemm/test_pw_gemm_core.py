"""Float64 check of the wave-specialised pointwise forward core (tools/experimental/pw_gemm/pw_gemm.hip).  Needs a library
built with AVDF_PW_EXPERIMENTAL=1; run by path:  python -m pytest tools/experimental/pw_gemm/test_pw_gemm_core.py"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests"))
from test_gpu_ops import rnd, mk_mask, g32, close, ops  # noqa: E402,F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 334, 256, 328), (2, 1536, 192, 256), (1, 200, 512, 64)])
@pytest.mark.parametrize("cfg,wpc", [(0, 0), (0, 1), (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (6, 0), (7, 0), (8, 0), (9, 0), (10, 0)])
def test_pipelined_pointwise_gemm_core(ops, cfg, wpc, shape):
    """The wave-specialised LDS-DMA forward core of tools/experimental/pw_gemm (only in a library built with
    AVDF_PW_EXPERIMENTAL=1: it is not part of the product build; plan codes >= 8, forced through the tuner hook):
    ragged and interior M / N tiles, one and several tiles per workgroup, 6 / 8 / 16 K-steps, bias, GELU with the pre-activation copy, row mask, and the fused
    AffineDropPath + residual epilogue, against float64."""
    import ctypes
    from audio_visual_deepfake_detection_amd._lib import lib, check, LIB_PATH
    raw = ctypes.CDLL(LIB_PATH)
    if not hasattr(raw, "avdf_pw_launch_count"):
        pytest.skip("product build: the experimental pointwise core is not compiled in (AVDF_PW_EXPERIMENTAL=1 builds it)")
    raw.avdf_pw_launch_count.restype = ctypes.c_int
    B, T, Cin, Cout = shape                       # (3, 334, ., 328): M = 1002 rows and 328 columns leave partial tiles
    x, w, b = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, 1, seed=2, scale=Cin ** -0.5), rnd(Cout, seed=3)
    m = mk_mask(B, T, [T, T - 40, 7][:B])
    pre = F.conv1d(x.transpose(1, 2), w, b).transpose(1, 2)
    ref = F.gelu(pre) * m[..., None]
    a, scale = rnd(B, T, Cout, seed=4), rnd(1, Cout, 1, seed=5, scale=0.3)
    keep = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9][:B]).double()
    ref_res = a + scale.view(1, 1, Cout) * keep.view(B, 1, 1) * ((pre) * m[..., None])
    check(lib.avdf_gemm_override(0, 8 + cfg, wpc), "override")
    n0 = raw.avdf_pw_launch_count()
    try:
        xg = g32(x, True)
        y = ops.conv1d(xg, g32(w), g32(b), g32(m), 1, ops.ACT_GELU)
        out = ops.conv1d_residual(g32(x), g32(w), g32(b), g32(m), g32(a), g32(scale), g32(keep))
        torch.cuda.synchronize()
    finally:
        check(lib.avdf_gemm_override(0, -1, -1), "override")
    assert raw.avdf_pw_launch_count() == n0 + 2, "the forced plan did not reach the pointwise core"
    close(y, ref, f"pw core cfg {cfg} wpc {wpc}: gelu(conv) * mask")
    close(out, ref_res, f"pw core cfg {cfg} wpc {wpc}: residual epilogue")
    # the pre-activation copy feeds the backward: its data gradient must match float64 as well
    gy = rnd(B, T, Cout, seed=6)
    xr = x.clone().requires_grad_(True)
    (F.gelu(F.conv1d(xr.transpose(1, 2), w, b).transpose(1, 2)) * m[..., None] * gy).sum().backward()
    y.backward(g32(gy))
    close(xg.grad, xr.grad, f"pw core cfg {cfg}: dgrad through the saved pre-activation")
