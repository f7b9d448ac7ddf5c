"""Pointwise-GEMM core (csrc/pw_gemm.hip) against the table plans of gemm_conv.hip on the K = 1 shapes of the exp10 step:
correctness against a float64 reference, then HIP-event time per launch of every tile configuration (hipGraph of 20
back-to-back launches, L2-warm).

    python tools/pwgemm_bench.py [fwd|bwd] [--quick]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from audio_visual_deepfake_detection_amd._lib import lib, check
from audio_visual_deepfake_detection_amd._lib import LIB_PATH
from tools.gemm_bench import timeit
import ctypes

RAW = ctypes.CDLL(LIB_PATH)
RAW.avdf_pw_launch_count.restype = ctypes.c_int

SHAPES = [  # name, M, Cin, Cout, nprob, act, has_mask, residual
    ("fc1 768", 6144, 256, 1024, 1, 2, False, False),
    ("fc2 768", 6144, 1024, 256, 1, 0, True, True),
    ("proj 768", 6144, 256, 256, 1, 0, True, False),
    ("qkv 768", 6144, 256, 256, 3, 0, False, False),
    ("fc1 384", 3072, 256, 1024, 1, 2, False, False),
    ("fc2 384", 3072, 1024, 256, 1, 0, True, True),
    ("qkv 384", 3072, 256, 256, 3, 0, False, False),
    ("proj 192", 1536, 256, 256, 1, 0, True, False),
    ("fc1 96", 768, 256, 1024, 1, 2, False, False),
]
NCFG = 11


def ptr_table(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def fwd_case(name, M, Cin, Cout, nprob, act, has_mask, residual, quick):
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cpu").manual_seed(1)
    xs = [torch.randn(1, M, Cin, generator=g).to(dev) for _ in range(nprob)]
    ws = [(torch.randn(Cout, Cin, 1, generator=g) * Cin ** -0.5).to(dev) for _ in range(nprob)]
    bs = [torch.randn(Cout, generator=g).to(dev) for _ in range(nprob)]
    mask = (torch.rand(1, M, generator=g) > 0.2).float().to(dev) if has_mask else None
    res = torch.randn(1, M, Cout, generator=g).to(dev) if residual else None
    rscale = torch.randn(Cout, generator=g).to(dev) if residual else None
    rkeep = (torch.rand(8, generator=g) > 0.3).float().to(dev) if residual else None   # 8 drop-path groups of M/8 rows
    ys = [torch.empty(1, M, Cout, device=dev) for _ in range(nprob)]
    ypre = torch.empty(1, M, Cout, device=dev) if act else None
    out = torch.empty(1, M, Cout, device=dev) if residual else None

    def launch():
        if residual:
            check(lib.avdf_conv1d_fwd_residual(xs[0].data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), mask.data_ptr(),
                                               res.data_ptr(), rscale.data_ptr(), rkeep.data_ptr(), ys[0].data_ptr(),
                                               out.data_ptr(), 8, M // 8, Cin, M // 8, Cout, 1, 1, st()), "fwd_res")
        elif nprob == 3:
            check(lib.avdf_conv1d_fwd_grouped(3, ptr_table(xs), ptr_table(ws), ptr_table(bs), None, ptr_table(ys), None, 1,
                                              M, Cin, M, Cout, 1, 1, act, st()), "fwd3")
        else:
            check(lib.avdf_conv1d_fwd(xs[0].data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(),
                                      mask.data_ptr() if has_mask else None, ys[0].data_ptr(),
                                      ypre.data_ptr() if act else None, 1, M, Cin, M, Cout, 1, 1, act, st()), "fwd")

    # float64 reference
    refs = []
    for i in range(nprob):
        pre = xs[i][0].double().cpu() @ ws[i][:, :, 0].double().cpu().t() + bs[i].double().cpu()
        y = torch.nn.functional.gelu(pre) if act == 2 else pre
        if has_mask:
            y = y * mask[0].double().cpu()[:, None]
        refs.append((pre, y))
    if residual:
        kp = rkeep.double().cpu().repeat_interleave(M // 8)
        ref_out = res[0].double().cpu() + rscale.double().cpu()[None, :] * kp[:, None] * refs[0][1]
    fl = 2.0 * M * Cin * Cout * nprob
    check(lib.avdf_gemm_override(0, -1, -1), "ovr")
    t_old = timeit(launch)
    line = f"{name:9s} GF={fl / 1e9:5.2f} peak={fl / 157.3e6:5.1f}us  table {t_old:6.1f}us ({fl / t_old / 1e6:5.1f}TF) |"
    best = (1e9, None)
    for cfg in range(NCFG):
        for stages in ((0, 1) if not quick else (0,)):    # workgroups per CU the persistent grid is sized for (0 = auto)
            check(lib.avdf_gemm_override(0, 8 + cfg, stages), "ovr")
            for y in ys:
                y.zero_()
            n0 = RAW.avdf_pw_launch_count()
            launch()
            torch.cuda.synchronize()
            if RAW.avdf_pw_launch_count() == n0:
                continue                                  # configuration not applicable to this shape: the table plan ran
            err = 0.0
            for i in range(nprob):
                sc = float(refs[i][1].abs().max())
                err = max(err, float((ys[i][0].double().cpu() - refs[i][1]).abs().max()) / sc)
            if act:
                err = max(err, float((ypre[0].double().cpu() - refs[0][0]).abs().max()) / float(refs[0][0].abs().max()))
            if residual:
                err = max(err, float((out[0].double().cpu() - ref_out).abs().max()) / float(ref_out.abs().max()))
            if err >= 2e-5:
                line += f" c{cfg}w{stages} WRONG({err:.1e})"
                continue
            t = timeit(launch)
            if t < best[0]:
                best = (t, (cfg, stages))
            line += f" c{cfg}w{stages} {t:5.1f}"
    check(lib.avdf_gemm_override(0, -1, -1), "ovr")
    print(line)
    print(f"{'':9s} best v2 {best[1]} {best[0]:6.1f}us = {fl / best[0] / 1e6:5.1f} TF/s ({fl / best[0] / 157.3e6 * 100:4.1f}% of peak), "
          f"{t_old / best[0]:4.2f}x the table plan", flush=True)


def main():
    quick = "--quick" in sys.argv
    shapes = SHAPES
    if "--b32" in sys.argv:            # the same layers at 32 clips per step
        shapes = [(n + " x4", 4 * M, ci, co, p, a, hm, r) for (n, M, ci, co, p, a, hm, r) in SHAPES[:4]]
    for sh in shapes:
        fwd_case(*sh, quick)


if __name__ == "__main__" and "--ablate" not in sys.argv and "--clock" not in sys.argv:
    main()


def ablate():
    """Where the time of a forward launch goes: the kernel's diagnostic switches (avdf_pw_flags) take out the
    epilogue items, the operand DMAs or the priority raise of the MFMA waves, one at a time and combined."""
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream
    RAW.avdf_pw_flags.argtypes = [ctypes.c_int]
    for name, M, Cin, Cout, cfgs in (("fc1 768", 6144, 256, 1024, (0, 6, 4, 8)), ("fc2 768", 6144, 1024, 256, (0, 6, 1))):
        x = torch.randn(1, M, Cin, device=dev)
        w = torch.randn(Cout, Cin, 1, device=dev) * Cin ** -0.5
        b = torch.randn(Cout, device=dev)
        y = torch.empty(1, M, Cout, device=dev)
        ypre = torch.empty(1, M, Cout, device=dev)
        fl = 2.0 * M * Cin * Cout
        act = 2 if Cout == 1024 else 0
        fn = lambda: check(lib.avdf_conv1d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(),
                                               ypre.data_ptr() if act else None, 1, M, Cin, M, Cout, 1, 1, act, st()), "fwd")
        for cfg in cfgs:
            for wpc in (0, 1):
                check(lib.avdf_gemm_override(0, 8 + cfg, wpc), "ovr")
                res = []
                for flags in (0, 16, 2, 4, 6):
                    assert RAW.avdf_pw_flags(flags) == 0
                    res.append((flags, timeit(fn)))
                RAW.avdf_pw_flags(0)
                names = {0: "full", 16: "epilogue reads only", 2: "no epilogue", 4: "no DMA", 6: "MFMA waves only"}
                print(f"{name} cfg {cfg} wpc {wpc} (MFMA peak {fl / 157.3e6:4.1f} us): " +
                      "  ".join(f"{names[f]} {t:5.1f}" for f, t in res), flush=True)
    check(lib.avdf_gemm_override(0, -1, -1), "ovr")


if __name__ == "__main__" and "--ablate" in sys.argv:
    ablate()


def clock():
    """In-kernel shader clock and workgroup lifetimes of the forward core (s_memtime / s_memrealtime stamps)."""
    import ctypes
    import numpy as np
    from audio_visual_deepfake_detection_amd import _lib
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.avdf_pw_stamps.argtypes = [ctypes.c_void_p]
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream
    stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for name, M, Cin, Cout, cfg, stages, act in (("fc1 768", 6144, 256, 1024, 0, 0, 2), ("fc1 768", 6144, 256, 1024, 6, 0, 2),
                                                 ("fc1 768", 6144, 256, 1024, 4, 0, 2), ("fc1 768", 6144, 256, 1024, 8, 0, 2),
                                                 ("proj 768", 6144, 256, 256, 0, 0, 0), ("proj 768", 6144, 256, 256, 2, 0, 0),
                                                 ("fc2 768", 6144, 1024, 256, 0, 0, 0), ("fc2 768", 6144, 1024, 256, 2, 0, 0)):
        x = torch.randn(1, M, Cin, device=dev)
        w = torch.randn(Cout, Cin, 1, device=dev) * Cin ** -0.5
        b = torch.randn(Cout, device=dev)
        y = torch.empty(1, M, Cout, device=dev)
        ypre = torch.empty(1, M, Cout, device=dev)
        check(lib.avdf_gemm_override(0, 8 + cfg, stages), "ovr")
        fn = lambda: check(lib.avdf_conv1d_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(),
                                               ypre.data_ptr() if act else None, 1, M, Cin, M, Cout, 1, 1, act, st()), "fwd")
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        assert raw.avdf_pw_stamps(ctypes.c_void_p(stamps.data_ptr())) == 0
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        raw.avdf_pw_stamps(None)
        s_all = stamps.cpu().numpy().reshape(-1, 8)
        ng = int((s_all[:, 3] > 0).sum())                         # MFMA-wave rows, then as many loader rows and epilogue rows
        s, mw, ew = s_all[:ng], s_all[ng:2 * ng, :3].astype(float), s_all[2 * ng:3 * ng, :2].astype(float)
        clk = (s[:, 1] - s[:, 0]) / np.maximum(s[:, 3] - s[:, 2], 1) * 100.0       # MHz
        life = (s[:, 3] - s[:, 2]) / 100.0                                           # us
        span = (s[:, 3].max() - s[:, 2].min()) / 100.0
        print(f"{name} cfg {cfg} wpc {stages} act {act}: {len(s)} workgroups, clock median {np.median(clk):.0f} MHz "
              f"(p10 {np.percentile(clk, 10):.0f}, p90 {np.percentile(clk, 90):.0f}); workgroup life median {np.median(life):.1f} us, "
              f"max {life.max():.1f}; first start -> last end {span:.1f} us", flush=True)
        tot = (s[:, 1] - s[:, 0]).astype(float)
        nsteps = -(-(M // (128 if cfg in (4, 7, 8) else 64)) * (Cout // (32 if cfg == 1 else 64)) // ng) * (Cin // 32)
        print("    MFMA wave 0, median share of the workgroup life: lds wait+barrier %.0f%%  reads+MFMA %.0f%%  reduce+park %.0f%%"
              % tuple(100 * np.median(s[:, i] / tot) for i in (4, 6, 7)), flush=True)
        print("    loader wave 0: data wait %.0f%%  barrier %.0f%%  DMA issue %.0f%% (cycles per step %s);  epilogue wave 0: barrier %.0f%%  work %.0f%% (%s)"
              % (*[100 * np.median(mw[:, i] / tot) for i in range(3)], " / ".join("%.0f" % (np.median(mw[:, i]) / max(nsteps, 1)) for i in range(3)),
                 *[100 * np.median(ew[:, i] / tot) for i in range(2)], " / ".join("%.0f" % (np.median(ew[:, i]) / max(nsteps, 1)) for i in range(2))),
              flush=True)
        stamps.zero_()
    check(lib.avdf_gemm_override(0, -1, -1), "ovr")


if __name__ == "__main__" and "--clock" in sys.argv:
    clock()
