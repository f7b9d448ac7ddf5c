"""Numerics + speed of the split-bf16 (bf16x3) forward GEMM mode against the exact-fp32 mode.  python tools/bf3_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check
from tools.gemm_bench import timeit, SHAPES

dev = torch.device("cuda:0")
st = lambda: torch.cuda.current_stream().cuda_stream
for name, B, T, Cin, Cout, K, s in SHAPES:
    Tout = (T + 2 * (K // 2) - K) // s + 1
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Cin, generator=g); w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    ref = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), None, s, K // 2).transpose(1, 2)
    xd, wd = x.to(dev), w.to(dev)
    y = torch.empty(B, Tout, Cout, device=dev)
    fl = 2.0 * B * Tout * Cout * Cin * K
    out = []
    for mode in (0, 1):
        check(lib.avdf_gemm_mode(mode), "mode")
        fn = lambda: check(lib.avdf_conv1d_fwd(xd.data_ptr(), wd.data_ptr(), None, None, y.data_ptr(), None, B, T, Cin, Tout, Cout, K, s, 0, st()), "fwd")
        fn(); torch.cuda.synchronize()
        err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
        us = timeit(fn)
        out.append((err, us))
    check(lib.avdf_gemm_mode(0), "mode")
    print(f"{name:14s} GF={fl/1e9:6.2f} | fp32: err {out[0][0]:.2e} {out[0][1]:7.1f} us {fl/out[0][1]/1e6:6.1f} TF | bf16x3: err {out[1][0]:.2e} {out[1][1]:7.1f} us {fl/out[1][1]/1e6:6.1f} TF", flush=True)
