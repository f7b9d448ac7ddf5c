"""The K = 1 backward launches of the LOWER pyramid levels (M = 8 x 24 ... 8 x 384 token rows): fused pair launch against
its data-gradient and weight-gradient halves alone and against an empty kernel -- how much of the ~15 us per launch is
launch floor, how much each half.   python tools/small_pair_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check
from tools.gemm_bench import timeit

dev = torch.device("cuda:0")
B = 8
tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
st = lambda: torch.cuda.current_stream().cuda_stream
tiny = torch.zeros(64, 256, device=dev); g = torch.ones(256, device=dev); bt = torch.zeros(256, device=dev); out = torch.empty_like(tiny)
floor = timeit(lambda: check(lib.avdf_layernorm_fwd(tiny.data_ptr(), g.data_ptr(), bt.data_ptr(), out.data_ptr(), None, None,
                                                    64, 256, 1e-5, 0, None, None, 64, st()), "ln"))
print(f"launch floor (LayerNorm of 64 rows, graph replay): {floor:.1f} us")
for T in (24, 48, 96, 192, 384, 768):
    for Cin, Cout in ((256, 256), (256, 1024), (1024, 256)):
        x = torch.randn(B, T, Cin, device=dev); w = torch.randn(Cout, Cin, 1, device=dev); dy = torch.randn(B, T, Cout, device=dev)
        dx = torch.empty(B, T, Cin, device=dev); dw = torch.zeros(Cout, Cin, 1, device=dev); db = torch.zeros(Cout, device=dev)
        ws = torch.empty(int(lib.avdf_conv1d_bwd_weight_ws(B, T, Cin, Cout, 1)), device=dev)
        pair = lambda: check(lib.avdf_conv1d_bwd_pair_grouped(1, tab([dy]), tab([w]), tab([x]), None, None, tab([dx]), tab([dw]), tab([db]),
                                                              B, T, Cin, Cout, 0, st()), "pair")
        dg = lambda: check(lib.avdf_conv1d_bwd_data(dy.data_ptr(), w.data_ptr(), None, None, dx.data_ptr(), B, T, Cin, T, Cout, 1, 1, 0, st()), "dgrad")
        wg = lambda: check(lib.avdf_conv1d_bwd_weight(dy.data_ptr(), x.data_ptr(), None, None, dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                                      B, T, Cin, T, Cout, 1, 1, 0, 1, st()), "wgrad")
        fw = lambda: check(lib.avdf_conv1d_fwd(x.data_ptr(), w.data_ptr(), None, None, dy.data_ptr(), None, B, T, Cin, T, Cout, 1, 1, 0, st()), "fwd")
        fl = 2.0 * B * T * Cin * Cout
        print(f"M={B * T:5d} {Cin:4d}->{Cout:4d}: pair {timeit(pair):6.1f} us  dgrad alone {timeit(dg):6.1f}  wgrad alone {timeit(wg):6.1f}  "
              f"forward {timeit(fw):6.1f}   (MFMA time at peak: fwd {fl / 157.3e6:4.2f} us, pair {2 * fl / 157.3e6:4.2f} us)", flush=True)
