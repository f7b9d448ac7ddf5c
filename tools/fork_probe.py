"""Does a hipGraph with a forked branch overlap a chain of tiny dependent kernels (a lower pyramid level) with a few
medium kernels (the query side of the next full-resolution cross-attention block)?  Sequential capture vs fork / join
capture of the same launches.   python tools/fork_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check

dev = torch.device("cuda:0")
g = torch.ones(256, device=dev); b = torch.zeros(256, device=dev)
small = [torch.randn(192, 256, device=dev) for _ in range(2)]
big = [torch.randn(6144, 256, device=dev) for _ in range(2)]
w = torch.randn(256, 256, 1, device=dev) * 0.06
yb = torch.empty(1, 6144, 256, device=dev)


def ln(x, y, M):
    check(lib.avdf_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, M, 256, 1e-5, 0, None, None, M,
                                 torch.cuda.current_stream().cuda_stream), "ln")


def small_chain(n):
    for i in range(n):
        ln(small[i & 1], small[1 - (i & 1)], 192)


def big_side():
    ln(big[0], big[1], 6144)
    ln(big[1], big[0], 6144)
    check(lib.avdf_conv1d_fwd(big[0].data_ptr(), w.data_ptr(), None, None, yb.data_ptr(), None, 1, 6144, 256, 6144, 256, 1, 1, 0,
                              torch.cuda.current_stream().cuda_stream), "fwd")


def timeit(gr, n=200):
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        gr.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s0):
    small_chain(4); big_side()
torch.cuda.synchronize()
for nsmall in (25, 50):
    for reps in (1, 5):
        gs = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gs, stream=s0):
            for _ in range(reps):
                small_chain(nsmall); big_side()
        gf = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf, stream=s0):
            for _ in range(reps):
                s1.wait_stream(s0)
                with torch.cuda.stream(s1):
                    big_side()
                small_chain(nsmall)
                s0.wait_stream(s1)
        go = torch.cuda.CUDAGraph()
        with torch.cuda.graph(go, stream=s0):
            for _ in range(reps):
                small_chain(nsmall)
        gb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gb, stream=s0):
            for _ in range(reps):
                big_side()
        print(f"{reps} x ({nsmall} tiny dependent launches + 3 medium): sequential {timeit(gs) / reps:7.1f} us  forked {timeit(gf) / reps:7.1f} us  "
              f"(tiny chain alone {timeit(go) / reps:6.1f}, medium alone {timeit(gb) / reps:6.1f})", flush=True)
