"""Time the C3D visual encoder on one 768-frame clip (2 chunks) and per layer.  python tools/c3d_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd.extractors import C3DVideoEncoder, video_features
from audio_visual_deepfake_detection_amd._lib import lib, check
from audio_visual_deepfake_detection_amd import ops

torch.manual_seed(0)
m = C3DVideoEncoder().cuda().pack()
fr = torch.randint(0, 256, (768, 224, 224, 3), dtype=torch.uint8, device="cuda")
for trim in (True, False):
    video_features(fr, m, trim_tail=trim); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        y = video_features(fr, m, trim_tail=trim)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 1.77e12 * (512 + (264 if trim else 512)) / 512
    print(f"trim_tail={trim}: {dt*1e3:.1f} ms per 768-frame clip, {fl/dt/1e12:.1f} TFLOP/s, {1/dt:.2f} clips/s", flush=True)
# per-layer timing on one 512-frame chunk (AVDF_C3D_BF3=1: split-bf16 operands for the Cin >= 16 layers)
bf3 = int(os.environ.get("AVDF_C3D_BF3", "0"))
check(lib.avdf_conv3d_mode(bf3), "conv3d_mode")
print(f"per layer, {'split-bf16' if bf3 else 'exact fp32'}:")
st = torch.cuda.current_stream().cuda_stream
T = 512
shapes = [(4, 64, 96), (64, 64, 96), (64, 96, 32), (96, 96, 32), (96, 128, 16), (128, 128, 16), (128, 128, 8), (128, 128, 4)]
for cin, cout, s in shapes:
    x = torch.randn(1, T, s, s, cin, device="cuda"); w = torch.randn(27, cout, cin, device="cuda"); b = torch.zeros(cout, device="cuda")
    y = torch.empty(1, T, s, s, cout, device="cuda")
    f = lambda: check(lib.avdf_conv3d_k3(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, T, s, s, cin, cout, 3, st), "c")
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl = 2.0 * T * s * s * 27 * (3 if cin == 4 else cin) * cout
    print(f"conv3d {cin:3d}->{cout:3d} @{s:2d}x{s:2d}: {ms:8.3f} ms  {fl/ms/1e9:6.1f} TFLOP/s", flush=True)
