"""Measure every GEMM plan for the conv shapes of one exp10 training step and write csrc/gemm_tuned.inc.

    python tools/autotune.py [--batch 8] [--out audio_visual_deepfake_detection_amd/csrc/gemm_tuned.inc]

1. runs one eager training step with ops.shape_log on -> the set of (kind, K, stride, B, T, Cin, Cout, nprob) shapes;
2. for every shape times all (tile, K-groups) choices (fwd / dgrad) or (token splits, K-groups) (wgrad) through
   avdf_gemm_override, `iters` launches captured in one hipGraph (L2-warm steady state, no host gaps);
3. keeps a tuned entry only where it beats the heuristic plan by > 3 %, writes the table and a JSON report.
Rebuild the library afterwards (python audio_visual_deepfake_detection_amd/csrc/build.py)."""
import os, argparse, json, os, sys
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
    if kind == "fwd":
        fn = lambda: check(lib.avdf_conv1d_fwd_grouped(n, T(xs), T(ws_), None, None, T(ys), None, B, Tin, Cin, Tout, Cout, K, s, 0, st()), "f")
        cands = [(t, kg + 8 * db) for t in range(5) for kg in (1, 2, 4) if kg <= (1, 2, 4, 4, 2)[t] for db in (0, 1)]
    elif kind == "dgrad":
        fn = lambda: check(lib.avdf_conv1d_bwd_data_grouped(n, T(ys), T(ws_), None, None, T(xs), B, Tin, Cin, Tout, Cout, K, s, 0, st()), "d")
        cands = [(t, kg + 8 * db) for t in range(5) for kg in (1, 2, 4) if kg <= (1, 2, 4, 4, 2)[t] for db in (0, 1)]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[8], help="clips per GPU per step; several values -> one merged table")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "audio_visual_deepfake_detection_amd", "csrc", "gemm_tuned.inc"))
    ap.add_argument("--report", default="gpurun_out/autotune.json")
    ap.add_argument("--bf16x3-fwd", action="store_true",
                    help="measure only the forward shapes, in split-bf16 mode, and APPEND them as kind-3 rows to --out")
    args = ap.parse_args()
    log = {}
    for bsz in args.batch:
        lg = collect(bsz)
        print(f"batch {bsz}: {len(lg)} distinct GEMM shapes, {sum(lg.values())} launches per step", flush=True)
        for k, v in lg.items():
            log.setdefault(k, v)
    rows, report, saved = [], [], 0.0
    if args.bf16x3_fwd:
        log = {k: v for k, v in log.items() if k[0] == "fwd"}
        check(lib.avdf_gemm_mode(1), "mode")
    for shape, cnt in sorted(log.items(), key=lambda kv: kv[0]):
        base, best, best_us, res = bench_shape(shape, args.iters)
        kind, K, s, B, Tin, Cin, Cout, n = shape
        Tout = (Tin + 2 * (K // 2) - K) // s + 1
        gain = (base - best_us) * cnt
        print(f"{shape} x{cnt}: heuristic {base:7.1f} us, best {best} {best_us:7.1f} us  (saves {gain:6.1f} us/step)", flush=True)
        report.append({"shape": list(shape), "launches": cnt, "heuristic_us": round(base, 2), "best": best, "best_us": round(best_us, 2), "all": res})
        if best is not None and best_us < 0.97 * base:
            saved += gain
            if kind == "fwd":
                key = (3 if args.bf16x3_fwd else 0, K, s, B * n * Tout, Cout, Cin)
            elif kind == "dgrad":
                key = (1, K, s, B * n * Tin, Cin, Cout)
            else:
                key = (2, K, s, B * Tout, Cout * n, Cin)
            rows.append("    {%d, %d, %d, %d, %d, %d, %d, %d},   // %s x%d: %.1f -> %.1f us" % (*key, best[0], best[1], kind, cnt, base, best_us))
    if args.bf16x3_fwd:
        check(lib.avdf_gemm_mode(0), "mode")
        cur = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "audio_visual_deepfake_detection_amd", "csrc", "gemm_tuned.inc")
        keep = [l for l in open(cur).read().splitlines() if not l.startswith("{3,") and "(kind 3)" not in l]
        with open(args.out, "w") as f:              # the table in the tree minus its old kind-3 rows, then the new ones
            f.write("\n".join(keep) + "\n")
            f.write("// forward shapes re-measured in split-bf16 mode (kind 3)\n" + "\n".join(rows) + ("\n" if rows else ""))
    else:
        with open(args.out, "w") as f:
            f.write("// generated by tools/autotune.py (B=%s clips, exp10) -- {kind, K, stride, M, N, Kred, a, b},\n" % ",".join(map(str, args.batch)))
            f.write("\n".join(rows) + ("\n" if rows else ""))
    os.makedirs(os.path.dirname(args.report) or ".", exist_ok=True)
    json.dump(report, open(args.report, "w"), indent=1)
    print(f"wrote {len(rows)} tuned plans to {args.out}; predicted saving {saved/1e3:.2f} ms/step")


if __name__ == "__main__":
    main()
