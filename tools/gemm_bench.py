"""Micro-benchmark of the MFMA conv kernels on the exp10 shapes (HIP events, L2-warm steady state).
    python tools/gemm_bench.py            # env AVDF_FORCE_TILE / AVDF_FORCE_KG override the heuristics
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check

SHAPES = [  # name, B, T, Cin, Cout, K, stride
    ("qkv/proj 768", 8, 768, 256, 256, 1, 1),
    ("fc1 768", 8, 768, 256, 1024, 1, 1),
    ("fc2 768", 8, 768, 1024, 256, 1, 1),
    ("qkv 384", 8, 384, 256, 256, 1, 1),
    ("qkv 96", 8, 96, 256, 256, 1, 1),
    ("embd0 k3", 8, 768, 3072, 256, 3, 1),
    ("head k3 768", 8, 768, 256, 256, 3, 1),
    ("down1 k3s2", 8, 768, 3072, 256, 3, 2),
]


def timeit(fn, iters=20):
    """GPU time per launch: `iters` launches captured in one hipGraph (no CPU launch gaps)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (5 * iters) * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    st_fn = lambda: torch.cuda.current_stream().cuda_stream
    print(f"tile={os.environ.get('AVDF_FORCE_TILE')} kg={os.environ.get('AVDF_FORCE_KG')}")
    for name, B, T, Cin, Cout, K, s in SHAPES:
        Tout = (T + 2 * (K // 2) - K) // s + 1
        x = torch.randn(B, T, Cin, device=dev)
        w = torch.randn(Cout, Cin, K, device=dev) * 0.05
        y = torch.empty(B, Tout, Cout, device=dev)
        dy = torch.randn(B, Tout, Cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        ws = torch.empty(int(lib.avdf_conv1d_bwd_weight_ws(B, Tout, Cin, Cout, K)), device=dev)
        fl = 2.0 * B * Tout * Cout * Cin * K
        tf = timeit(lambda: check(lib.avdf_conv1d_fwd(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, B, T,
                                                      Cin, Tout, Cout, K, s, 0, st_fn()), "fwd"))
        td = timeit(lambda: check(lib.avdf_conv1d_bwd_data(dy.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), B, T,
                                                           Cin, Tout, Cout, K, s, 0, st_fn()), "dgrad"))
        tw = timeit(lambda: check(lib.avdf_conv1d_bwd_weight(dy.data_ptr(), x.data_ptr(), None, None, dw.data_ptr(), None,
                                                             ws.data_ptr(), B, T, Cin, Tout, Cout, K, s, 0, 0, st_fn()), "wgrad"))
        print(f"{name:14s} GF={fl/1e9:6.2f}  fwd {tf:7.1f}us {fl/tf/1e6:6.1f}TF | dgrad {td:7.1f}us {fl/td/1e6:6.1f}TF | "
              f"wgrad+red {tw:7.1f}us {fl/tw/1e6:6.1f}TF")


if __name__ == "__main__":
    main()
