"""Per-extractor timing on one 768-frame clip (30.72 s of 16 kHz audio, 768x224x224 frames).  python tools/extract_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd.extractors import (LogMelFrontEnd, AudioNTT2020Task6, byola_features, Emotion2Vec,
                                                             emotion2vec_features, C3DVideoEncoder, video_features)
dev = torch.device("cuda:0")
torch.manual_seed(0)
fe, byola, e2v, c3d = LogMelFrontEnd().to(dev), AudioNTT2020Task6().to(dev).eval(), Emotion2Vec().to(dev), C3DVideoEncoder().to(dev)
pcm = torch.rand(491520, device=dev) - 0.5
frames = torch.randint(0, 256, (768, 224, 224, 3), device=dev, dtype=torch.uint8)


def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"BYOL-A (log-mel + CNN + FC): {t(lambda: byola_features(pcm, byola, fe)):.2f} ms")
print(f"Emotion2Vec:                 {t(lambda: emotion2vec_features(pcm, e2v)):.2f} ms")
print(f"C3D:                         {t(lambda: video_features(frames, c3d)):.2f} ms")
pcm8 = torch.rand(8, 491520, device=dev) - 0.5
print(f"Emotion2Vec, 8 clips batched: {t(lambda: e2v.forward_batch(pcm8)) / 8:.2f} ms per clip")
