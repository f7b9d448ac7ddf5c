"""K=1 backward pair (dgrad + accumulating wgrad in one launch) vs the two separate launches, per shape, and a sweep of
the pair's (dgrad tile, K-groups) x (wgrad token splits) plans.   python tools/pair_bench.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check
from tools.gemm_bench import timeit


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda:0")
    tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    shapes = []
    for T in (768, 384, 192, 96, 48, 24):
        shapes += [(3, T, 256, 256), (1, T, 256, 256), (1, T, 256, 1024), (1, T, 1024, 256)]
    shapes += [(1, 1512, 256, 256)]          # gapped head sequence
    for n, T, Cin, Cout in shapes:
        xs = [torch.randn(B, T, Cin, device=dev) for _ in range(n)]
        ws = [torch.randn(Cout, Cin, 1, device=dev) for _ in range(n)]
        dys = [torch.randn(B, T, Cout, device=dev) for _ in range(n)]
        dxs = [torch.empty(B, T, Cin, device=dev) for _ in range(n)]
        dws = [torch.zeros(Cout, Cin, 1, device=dev) for _ in range(n)]
        dbs = [torch.zeros(Cout, device=dev) for _ in range(n)]
        st = lambda: torch.cuda.current_stream().cuda_stream
        run = lambda: check(lib.avdf_conv1d_bwd_pair_grouped(n, tab(dys), tab(ws), tab(xs), None, None, tab(dxs), tab(dws), tab(dbs),
                                                             B, T, Cin, Cout, 0, st()), "pair")
        for k in range(3):
            check(lib.avdf_gemm_override(k, -1, -1), "ov")
        check(lib.avdf_bwd_pair_mode(0), "m")
        t_sep = timeit(run)
        check(lib.avdf_bwd_pair_mode(1), "m")
        t_pair = timeit(run)
        best = (t_pair, None)
        for tile in range(7):
            for kg in (1, 2, 4):
                if kg > (1, 2, 4, 4, 2, 4, 4)[tile]:
                    continue
                for splits in (1, 2, 4, 8, 16, 32, 64):
                    if splits > B * ((T + 127) // 128):
                        continue
                    check(lib.avdf_gemm_override(1, tile, kg), "ov")
                    check(lib.avdf_gemm_override(2, splits, kg), "ov")
                    t = timeit(run, iters=10)
                    if t < best[0]:
                        best = (t, (tile, kg, splits))
        print(f"n={n} B={B} T={T:5d} {Cin:4d}->{Cout:4d}: separate {t_sep:7.1f} us  pair(current plans) {t_pair:7.1f} us  "
              f"best pair {best[0]:7.1f} us {best[1]}", flush=True)
    for k in range(3):
        check(lib.avdf_gemm_override(k, -1, -1), "ov")


if __name__ == "__main__":
    main()
