"""Measure every GEMM plan for the conv shapes of one exp10 training step and write csrc/gemm_tuned.inc.

    python tools/autotune.py [--batch 8 32] [--kinds fwd dgrad wgrad fwd3 pair] [--out gpurun_out/gemm_tuned.inc]

1. runs one eager training step with ops.shape_log on -> the set of (kind, K, stride, B, T, Cin, Cout, nprob) shapes;
2. for every shape times all (tile, K-groups) choices (fwd / dgrad; fwd3 = the forward shapes again in split-bf16 mode),
   (token splits, K-groups) (wgrad) or (dgrad tile, K-groups, wgrad splits) (pair: the K=1 one-launch backward) through
   avdf_gemm_override, `iters` launches captured in one hipGraph (L2-warm steady state, no host gaps);
3. starts from the table in the tree, replaces every re-measured shape (an entry is kept only where it beats the baseline
   plan by > 3 %), writes the merged table to --out and a JSON report.
Copy --out over csrc/gemm_tuned.inc and rebuild (python audio_visual_deepfake_detection_amd/csrc/build.py)."""
import argparse, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from audio_visual_deepfake_detection_amd import build_model, ops
from audio_visual_deepfake_detection_amd._lib import lib, check
from audio_visual_deepfake_detection_amd.engine import TrainStep
from tools.gemm_bench import timeit

STREAMS = [(768, 256), (383, 2048), (1535, 768)]
KIND = {"fwd": 0, "dgrad": 1, "wgrad": 2}


def collect(batch):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model, cfg = build_model("deepfake_exp10.yaml")
    eng = TrainStep(model.to(dev), batch, STREAMS, use_graph=False)
    for b in range(batch):
        for dst in eng.streams[b]:
            dst.normal_()
        if b % 2 == 0:
            eng.gt[b, 0] = torch.tensor([100.0, 130.0]); eng.ngt[b] = 1
    eng.step()
    ops.shape_log = {}
    eng.step()
    torch.cuda.synchronize()
    log, ops.shape_log = ops.shape_log, None
    return log


def bench_shape(shape, iters):
    kind, K, s, B, Tin, Cin, Cout, n = shape
    dev = torch.device("cuda:0")
    Tout = (Tin + 2 * (K // 2) - K) // s + 1
    st = lambda: torch.cuda.current_stream().cuda_stream
    xs = [torch.randn(B, Tin, Cin, device=dev) for _ in range(n)]
    ws_ = [torch.randn(Cout, Cin, K, device=dev) * 0.05 for _ in range(n)]
    ys = [torch.randn(B, Tout, Cout, device=dev) for _ in range(n)]
    dws = [torch.zeros_like(w) for w in ws_]
    T = lambda ts: ops._ptr_table(ts)
    tiles = [(t, kg) for t in range(7) for kg in (1, 2, 4) if kg <= (1, 2, 4, 4, 2, 4, 4)[t]]
    if kind == "pair":
        # (dgrad tile, K-groups) x (wgrad token splits); the baseline is the table / heuristic plan of the two halves
        dxs = [torch.empty_like(x) for x in xs]
        fn = lambda: check(lib.avdf_conv1d_bwd_pair_grouped(n, T(ys), T(ws_), T(xs), None, None, T(dxs), T(dws), None, B, Tin, Cin, Cout, 0,
                                                            st()), "p")
        check(lib.avdf_gemm_override(4, -2, -2), "o")      # ignore kind-4 rows: the halves' own plans
        base = timeit(fn, iters)
        best, best_us, res = None, base, {}
        for t_, kg in tiles:
            for sp in (1, 2, 4, 8, 16, 32, 64):
                if sp > max(1, (B * Tin + 127) // 128):
                    continue
                check(lib.avdf_gemm_override(4, t_, kg | (sp << 3)), "o")
                us = timeit(fn, iters)
                res[f"{t_},{kg},{sp}"] = round(us, 2)
                if us < best_us:
                    best, best_us = (t_, kg | (sp << 3)), us
        check(lib.avdf_gemm_override(4, -1, -1), "o")
        return base, best, best_us, res
    if kind == "fwd":
        fn = lambda: check(lib.avdf_conv1d_fwd_grouped(n, T(xs), T(ws_), None, None, T(ys), None, B, Tin, Cin, Tout, Cout, K, s, 0, st()), "f")
        cands = [(t_, kg + 8 * db) for t_, kg in tiles for db in (0, 1)]
    elif kind == "dgrad":
        fn = lambda: check(lib.avdf_conv1d_bwd_data_grouped(n, T(ys), T(ws_), None, None, T(xs), B, Tin, Cin, Tout, Cout, K, s, 0, st()), "d")
        cands = [(t_, kg + 8 * db) for t_, kg in tiles for db in (0, 1)]
    else:
        wsb = torch.empty(8, device=dev)
        fn = lambda: check(lib.avdf_conv1d_bwd_weight_grouped(n, T(ys), T(xs), None, None, T(dws), None, wsb.data_ptr(), B, Tin, Cin, Tout,
                                                               Cout, K, s, 0, 1, st()), "w")
        cands = [(sp, kg) for sp in (1, 2, 4, 8, 16, 32, 64) for kg in (1, 2, 4)]
    k = KIND[kind]
    check(lib.avdf_gemm_override(k, -2, -2), "o")          # heuristic plan, table ignored
    base = timeit(fn, iters)
    best, best_us = None, base
    res = {}
    for a, b in cands:
        check(lib.avdf_gemm_override(k, a, b), "o")
        us = timeit(fn, iters)
        res[f"{a},{b}"] = round(us, 2)
        if us < best_us:
            best, best_us = (a, b), us
    check(lib.avdf_gemm_override(k, -1, -1), "o")
    return base, best, best_us, res


# forward GEMM shapes of the frozen extractors for one 768-frame clip (491520 samples): Emotion2Vec feature-extractor convs
# (valid, no padding), its transformer linears and per-head attention GEMMs, BYOL-A's fc layers.  (B, Tin, Cin, Tout, Cout, K, s)
EXTRACTOR_SHAPES = [(1, 98303, 512, 49151, 512, 3, 2), (1, 49151, 512, 24575, 512, 3, 2), (1, 24575, 512, 12287, 512, 3, 2),
                    (1, 12287, 512, 6143, 512, 3, 2), (1, 6143, 512, 3071, 512, 2, 2), (1, 3071, 512, 1535, 512, 2, 2),
                    (1, 1535, 512, 1535, 768, 1, 1), (1, 1535, 768, 1535, 2304, 1, 1), (1, 1535, 768, 1535, 768, 1, 1),
                    (1, 1535, 768, 1535, 3072, 1, 1), (1, 1535, 3072, 1535, 768, 1, 1),
                    (12, 1535, 64, 1535, 1535, 1, 1), (12, 1535, 1536, 1535, 64, 1, 1),
                    (1, 384, 512, 384, 2048, 1, 1), (1, 384, 2048, 384, 2048, 1, 1)]


def bench_ex(shape, iters):
    """One extractor forward shape through avdf_conv1d_fwd_strided (valid conv, natural strides) in split-bf16 mode."""
    B, Tin, Cin, Tout, Cout, K, s = shape
    dev = torch.device("cuda:0")
    x = torch.randn(B, Tin, Cin, device=dev)
    w = torch.randn(Cout, Cin * K, device=dev) * 0.05
    y = torch.empty(B, Tout, Cout, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream
    pad = 0
    assert Tout == (Tin + 2 * pad - K) // s + 1, shape
    fn = lambda: check(lib.avdf_conv1d_fwd_strided(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, Tin, Cin, Tout, Cout, K, s, pad,
                                                   Cin, Cin * K, Cout, Tin * Cin, 0, Tout * Cout, 0, st()), "ex")
    tiles = [(t, kg) for t in range(7) for kg in (1, 2, 4) if kg <= (1, 2, 4, 4, 2, 4, 4)[t]]
    if K != 1:
        tiles = [tk for tk in tiles if tk[0] != 5]
    check(lib.avdf_gemm_override(0, -2, -2), "o")
    base = timeit(fn, iters)
    best, best_us, res = None, base, {}
    for t_, kg in tiles:
        for db in (0, 1):
            check(lib.avdf_gemm_override(0, t_, kg + 8 * db), "o")
            us = timeit(fn, iters)
            res[f"{t_},{kg + 8 * db}"] = round(us, 2)
            if us < best_us:
                best, best_us = (t_, kg + 8 * db), us
    check(lib.avdf_gemm_override(0, -1, -1), "o")
    return base, best, best_us, res


TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audio_visual_deepfake_detection_amd", "csrc",
                     "gemm_tuned.inc")


def load_table(path):
    rows = {}
    for l in open(path).read().splitlines():
        m = re.match(r"\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},(.*)", l)
        if m:
            v = [int(x) for x in m.groups()[:8]]
            rows[tuple(v[:6])] = (v[6], v[7], m.group(9).strip())
    return rows


def write_table(path, table):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        f.write("// generated by tools/autotune.py (exp10 + extractor shapes) -- {kind, K, stride, M, N, Kred, a, b}; kinds: 0 fwd,\n"
                "// 1 dgrad, 2 wgrad (splits, kg), 3 fwd in split-bf16 mode, 4 K=1 backward pair (dgrad tile, kg | splits << 3)\n")
        for key in sorted(table):
            a, b, note = table[key]
            f.write("    {%d, %d, %d, %d, %d, %d, %d, %d},   %s\n" % (*key, a, b, note))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[8], help="clips per GPU per step; several values -> one merged table")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--kinds", nargs="+", default=["fwd", "dgrad", "wgrad", "fwd3", "pair"],
                    help="fwd / dgrad / wgrad: exact-fp32 kernels; fwd3: forward shapes in split-bf16 mode (table kind 3); "
                         "pair: K=1 dgrad+wgrad in one launch (table kind 4)")
    ap.add_argument("--out", default="gpurun_out/gemm_tuned.inc",
                    help="the table in the tree with every re-measured shape replaced; copy it over csrc/gemm_tuned.inc")
    ap.add_argument("--report", default="gpurun_out/autotune.json")
    ap.add_argument("--extractors", action="store_true",
                    help="measure only the frozen extractors' forward shapes (EXTRACTOR_SHAPES) in split-bf16 mode -> kind-3 rows")
    args = ap.parse_args()
    if args.extractors:
        table = load_table(TABLE)
        check(lib.avdf_gemm_mode(1), "mode")
        report = []
        for shape in EXTRACTOR_SHAPES:
            B, Tin, Cin, Tout, Cout, K, s = shape
            base, best, best_us, res = bench_ex(shape, args.iters)
            print(f"extractor fwd3 {shape}: heuristic {base:7.1f} us, best {best} {best_us:7.1f} us", flush=True)
            report.append({"mode": "fwd3-extractor", "shape": list(shape), "baseline_us": round(base, 2), "best": best,
                           "best_us": round(best_us, 2), "all": res})
            key = (3, K, s, B * Tout, Cout, Cin)
            table.pop(key, None)
            if best is not None and best_us < 0.97 * base:
                table[key] = (best[0], best[1], "// extractor fwd3: %.1f -> %.1f us" % (base, best_us))
        check(lib.avdf_gemm_mode(0), "mode")
        write_table(args.out, table)
        json.dump(report, open(args.report, "w"), indent=1)
        print(f"wrote {len(table)} tuned plans to {args.out}")
        return
    log = {}
    for bsz in args.batch:
        lg = collect(bsz)
        print(f"batch {bsz}: {len(lg)} distinct GEMM shapes, {sum(lg.values())} launches per step", flush=True)
        for k, v in lg.items():
            log.setdefault(k, v)
    table = load_table(TABLE)
    report, saved = [], 0.0
    work = []
    for shape, cnt in sorted(log.items(), key=lambda kv: kv[0]):
        if shape[0] in args.kinds:
            work.append((shape, cnt, shape[0]))
        if shape[0] == "fwd" and "fwd3" in args.kinds:
            work.append((shape, cnt, "fwd3"))
    for shape, cnt, mode in work:
        kind, K, s, B, Tin, Cin, Cout, n = shape
        Tout = (Tin + 2 * (K // 2) - K) // s + 1
        check(lib.avdf_gemm_mode(1 if mode == "fwd3" else 0), "mode")
        base, best, best_us, res = bench_shape(shape, args.iters)
        gain = (base - best_us) * cnt
        print(f"{mode} {shape[1:]} x{cnt}: baseline {base:7.1f} us, best {best} {best_us:7.1f} us  (saves {gain:6.1f} us/step)", flush=True)
        report.append({"mode": mode, "shape": list(shape), "launches": cnt, "baseline_us": round(base, 2), "best": best,
                       "best_us": round(best_us, 2), "all": res})
        if mode == "fwd" or mode == "fwd3":
            key = (3 if mode == "fwd3" else 0, K, s, B * n * Tout, Cout, Cin)
        elif mode == "dgrad":
            key = (1, K, s, B * n * Tin, Cin, Cout)
        elif mode == "pair":
            key = (4, K, s, B * n * Tin, Cin, Cout)
        else:
            key = (2, K, s, B * Tout, Cout * n, Cin)
        table.pop(key, None)
        if best is not None and best_us < 0.97 * base:
            saved += gain
            table[key] = (best[0], best[1], "// %s x%d: %.1f -> %.1f us" % (mode, cnt, base, best_us))
    check(lib.avdf_gemm_mode(0), "mode")
    write_table(args.out, table)
    os.makedirs(os.path.dirname(args.report) or ".", exist_ok=True)
    json.dump(report, open(args.report, "w"), indent=1)
    print(f"wrote {len(table)} tuned plans to {args.out}; predicted saving over the baselines {saved/1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
