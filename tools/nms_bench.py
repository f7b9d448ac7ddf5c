"""Where the time of the decode + NMS + voting launch goes (one workgroup per clip): full launch against the same launch
without voting, with a single kept segment, and with no candidate above the threshold.   python tools/nms_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd import ops
from tools.gemm_bench import timeit

dev = torch.device("cuda:0")
B, lens = 8, (768, 384, 192, 96, 48, 24)
P = sum(lens)
g = torch.Generator().manual_seed(0)
pts = []
for l, T in enumerate(lens):
    s = 2 ** l
    t = torch.arange(T, dtype=torch.float32) * s
    pts.append(torch.stack([t, torch.zeros(T), torch.full((T,), 1e4), torch.full((T,), float(s))], 1))
points = torch.cat(pts).to(dev)
# a trained detector's output: a few true segments per clip, every point near one votes for it with a high score
logits = torch.full((B, P), -6.0)
offsets = torch.rand(B, P, 2, generator=g) * 4 + 0.5
for b in range(B):
    for k in range(4):
        c, w = float(torch.randint(50, 700, (1,), generator=g)), float(torch.randint(10, 80, (1,), generator=g))
        near = (points[:, 0].cpu() - c).abs() < w
        logits[b, near] = torch.randn(int(near.sum()), generator=g) + 2.0
        st = points[near, 3].cpu()
        offsets[b, near, 0] = ((points[near, 0].cpu() - (c - w)) / st).clamp(min=0.1)
        offsets[b, near, 1] = (((c + w) - points[near, 0].cpu()) / st).clamp(min=0.1)
dense = torch.randn(B, P, generator=g)            # random-init model: every point a candidate
mask = torch.ones(B, P)
for name, lg in (("trained-like (4 segments per clip)", logits), ("random-init (bench.py --mode infer)", dense)):
    lg, of, mk = lg.to(dev), offsets.to(dev), mask.to(dev)
    run = lambda vt=0.75, ms=100, thr=0.001: ops.decode_nms(points, lg, of, mk, thr, 0.0, 0.001, 0.75, vt, ms)
    segs, scores, pidx, count = run()
    torch.cuda.synchronize()
    print(f"{name}: kept per clip {count.tolist()}")
    print(f"   full {timeit(run):6.1f} us   no voting {timeit(lambda: run(vt=0.0)):6.1f}   one kept segment {timeit(lambda: run(ms=1)):6.1f}"
          f"   no candidate {timeit(lambda: run(thr=2.0)):6.1f}", flush=True)
