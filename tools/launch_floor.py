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
