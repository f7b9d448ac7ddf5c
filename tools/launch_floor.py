"""Fixed cost of a dependent kernel launch on this stack: N trivial launches of one of the library's kernels
(scale_rows on 8 rows) -- eager back to back behind a GPU-side spin, and replayed from a hipGraph.

    python tools/launch_floor.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check

dev = torch.device("cuda:0")
x = torch.randn(8, 256, device=dev)
s = torch.ones(8, device=dev)
y = torch.empty_like(x)
st = lambda: torch.cuda.current_stream().cuda_stream
N = 400


def tiny():
    check(lib.avdf_scale_rows(x.data_ptr(), s.data_ptr(), y.data_ptr(), 8, 256, st()), "sr")


def ev_time(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def eager():
    torch.cuda._sleep(20_000_000)        # ~10 ms of GPU-side spin: the host queues everything meanwhile
    for _ in range(N):
        tiny()


for _ in range(3):
    tiny()
torch.cuda.synchronize()
spin = ev_time(lambda: torch.cuda._sleep(20_000_000))
t = ev_time(eager)
print(f"eager, queued behind a spin: {(t - spin) * 1e3 / N:.2f} us per launch (spin {spin:.2f} ms, total {t:.2f} ms)")

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        tiny()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(N):
        tiny()
g.replay()
torch.cuda.synchronize()
t = ev_time(g.replay)
print(f"hipGraph replay: {t * 1e3 / N:.2f} us per launch")
t0 = time.perf_counter()
for _ in range(N):
    tiny()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host cost of one ctypes launch: {(t1 - t0) * 1e6 / N:.2f} us")

# --- distinct kernels in rotation (cold instruction / kernarg caches per launch?) ---------------------------------
from audio_visual_deepfake_detection_amd import ops as O

u = torch.rand(8, device=dev)
kpv = torch.full((1,), 0.9, device=dev)
m8 = torch.ones(1, 64, device=dev)
xx = torch.randn(1, 64, 32, device=dev)
g1 = torch.ones(1, 32, 1, device=dev)
b1 = torch.zeros(1, 32, 1, device=dev)


def rotation():
    tiny()
    O.droppath_keep(u, kpv)
    O.mask_pyramid(m8, 3)
    O.layernorm(xx, g1, b1)
    O.maxpool3s2(xx)
    O.scale_rows(xx.view(-1, 32), m8.view(-1))


with torch.no_grad():
    rotation()
    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        rotation()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(60):
            rotation()
    g2.replay()
    torch.cuda.synchronize()
    t = ev_time(g2.replay)
    print(f"hipGraph replay, six distinct tiny kernels in rotation: {t * 1e3 / 360:.2f} us per launch")
    big = torch.empty(64 << 20, device=dev)          # 256 MB: evicts L2 (and most of the Infinity Cache) between rotations
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        for _ in range(20):
            big.zero_()
            rotation()
    g3.replay()
    torch.cuda.synchronize()
    t3 = ev_time(g3.replay)
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        for _ in range(20):
            big.zero_()
    g4.replay()
    torch.cuda.synchronize()
    t4 = ev_time(g4.replay)
    print(f"the same behind a 256 MB fill (cold L2): {(t3 - t4) * 1e3 / 120:.2f} us per launch (fill alone {t4 * 1e3 / 20:.1f} us)")

# --- a dependent chain of real small ops (each reads what the previous one wrote) ------------------------------------
with torch.no_grad():
    for rows in (64, 3072, 6144):
        z = torch.randn(1, rows, 256, device=dev)
        gw = torch.ones(1, 256, 1, device=dev)
        gb = torch.zeros(1, 256, 1, device=dev)
        mrow = torch.ones(rows, device=dev)

        def chain(z=z, gw=gw, gb=gb, mrow=mrow):
            t_ = z
            for _ in range(10):
                t_ = O.layernorm(t_, gw, gb)
                t_ = O.scale_rows(t_.view(-1, 256), mrow).view(1, -1, 256)
            return t_
        chain()
        torch.cuda.synchronize()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g5 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g5):
            for _ in range(10):
                chain()
        g5.replay()
        torch.cuda.synchronize()
        t5 = ev_time(g5.replay)
        print(f"dependent chain LN -> scale_rows on [{rows}, 256] ({rows * 1024 / 1e6:.2f} MB per tensor): {t5 * 1e3 / 200:.2f} us per launch")
