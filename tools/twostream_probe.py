"""Feasibility probe: do two independent half-batches captured as parallel branches of ONE hipGraph (two HIP streams) beat one
full batch?  Inference only (no shared gradient buffers).  python tools/twostream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd import build_model
from audio_visual_deepfake_detection_amd.engine import InferStep

STREAMS = [(768, 256), (383, 2048), (1535, 768)]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model, _ = build_model("deepfake_exp10.yaml")
model = model.to(dev).eval()


def fill(e):
    for buf in e.stream_bufs:
        buf.normal_()


def timeit(fn, n=60, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (8, 16):
    full = InferStep(model, B, STREAMS)
    fill(full)
    full.capture()
    t_full = timeit(full.step)
    h0, h1 = InferStep(model, B // 2, STREAMS, use_graph=False), InferStep(model, B // 2, STREAMS, use_graph=False)
    fill(h0); fill(h1)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        for _ in range(2):
            h0._run(); h1._run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s0):
        s1.wait_stream(s0)
        h0._run()
        with torch.cuda.stream(s1):
            h1._run()
        s0.wait_stream(s1)
    t_two = timeit(g.replay)
    # sequential halves in one graph (same launches, no concurrency): isolates the effect of the second stream
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s0):
        h0._run(); h1._run()
    t_seq = timeit(g2.replay)
    print(f"B={B}: one batch {t_full:.3f} ms | two halves, two streams {t_two:.3f} ms | two halves, one stream {t_seq:.3f} ms", flush=True)
