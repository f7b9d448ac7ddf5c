"""In-process, in-situ GEMM plan search: ONE process builds the model once, and for every candidate plan row sets it
through avdf_tuned_extra, re-captures the training step (hipGraph) and times it.  ~3 s per candidate instead of the
~12 s of tools/plan_ab.py (a bench.py process per candidate).

    python tools/plan_search.py --bf3 [--steps 40] [--batch 8]     (on the GPU box; writes gpurun_out/plan_search.json)

--bf3: the split-bf16 training mode (TrainStep(fwd_bf16x3, bwd_bf16x3)); rows searched are kind 5 (fused K=1 backward
in that mode) and kind 6 (data gradient in that mode).  Coordinate descent over the heaviest shapes."""
import argparse, copy, ctypes, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from audio_visual_deepfake_detection_amd import build_model
from audio_visual_deepfake_detection_amd._lib import lib, check
from audio_visual_deepfake_detection_amd.engine import TrainStep

STREAMS = [(768, 256), (383, 2048), (1535, 768)]
MAXKG = (1, 2, 4, 4, 2, 4, 4)


TILES = [1, 2, 3, 5, 6]         # --tiles; 0 = 128x128 and 4 = 256x64 (2x2 accumulator tiles per wave) are large-batch candidates


def pair_cands(splits=(4, 8, 16)):
    return [(t, kg | (sp << 3)) for t in TILES for kg in (1, 2, 4) if kg <= MAXKG[t] and not (t == 6 and kg == 1)
            for sp in splits]


def tile_cands():
    """data-gradient plans: (tile, K-groups [+ 8: double-buffered operand tiles])"""
    return [(t, kg + 8 * db) for t in TILES if t != 5 for kg in (1, 2, 4) if kg <= MAXKG[t] for db in (0, 1)]


def wgrad_cands():
    return [(sp, kg) for sp in (1, 2, 4, 8, 16, 32) for kg in (1, 2)]


def fwd_cands():
    return [(t, kg + 8 * db) for t in TILES if t != 5 for kg in (1, 2, 4) if kg <= MAXKG[t] for db in (0, 1)]


def shapes(batch, bf3, fwd=False):
    M = batch * 768
    pk, dk = (5, 6) if bf3 else (4, 1)
    if fwd:                 # forward K=1 / k=3 classes (kind 3 = forward in split-bf16 mode)
        fk = 3 if bf3 else 0
        return [((fk, 1, 1, M, 1024, 256), fwd_cands), ((fk, 1, 1, M, 256, 1024), fwd_cands), ((fk, 1, 1, 3 * M, 256, 256), fwd_cands),
                ((fk, 1, 1, M, 256, 256), fwd_cands), ((fk, 3, 1, M, 256, 256), fwd_cands), ((fk, 3, 2, M // 2, 256, 256), fwd_cands),
                ((fk, 1, 1, M // 2, 1024, 256), fwd_cands), ((fk, 1, 1, M // 2, 256, 1024), fwd_cands)]
    return [((pk, 1, 1, M, 256, 256), pair_cands), ((pk, 1, 1, 3 * M, 256, 256), pair_cands), ((pk, 1, 1, M, 1024, 256), pair_cands),
            ((pk, 1, 1, M, 256, 1024), pair_cands), ((pk, 1, 1, M // 2, 256, 256), pair_cands), ((pk, 1, 1, 3 * M // 2, 256, 256), pair_cands),
            ((dk, 3, 1, M, 256, 256), tile_cands), ((dk, 3, 2, M, 256, 256), tile_cands)]


def ranked_shapes(model0, batch, bf3, dev):
    """Table keys of every GEMM launch of one eager step, heaviest first: [(key, candidate fn, ms per step)]."""
    from audio_visual_deepfake_detection_amd import ops
    model = copy.deepcopy(model0).to(dev)
    eng = TrainStep(model, batch, STREAMS, lr=1e-4, use_graph=False, fwd_bf16x3=bf3, bwd_bf16x3=bf3)
    for b in range(batch):
        for dst in eng.streams[b]:
            dst.normal_()
        eng.ngt[b] = 0
    eng.gt.zero_()
    eng.gt[0, 0, 0], eng.gt[0, 0, 1], eng.ngt[0] = 100.0, 260.0, 1
    eng.step()
    torch.cuda.synchronize()
    ops.profile_begin()
    eng.step()
    torch.cuda.synchronize()
    recs = ops.profile_end()
    agg = {}
    for tag, fl, s, e, shape in recs:
        if shape is None:
            continue
        kind, K, st, B, Tin, Cin, Cout, n = shape
        Tout = (Tin + 2 * (K // 2) - K) // st + 1
        if kind == "fwd":
            key, c = ((3 if bf3 else 0), K, st, B * n * Tout, Cout, Cin), fwd_cands
        elif kind == "dgrad":
            key, c = ((6 if bf3 else 1), K, st, B * n * Tin, Cin, Cout), tile_cands
        elif kind == "pair":
            key, c = ((5 if bf3 else 4), 1, 1, B * n * Tin, Cin, Cout), pair_cands
        elif kind == "wgrad" and K == 3:
            key, c = (2, K, st, B * Tout, Cout * n, Cin), wgrad_cands
        else:
            continue
        a = agg.setdefault(key, [c, 0.0])
        a[1] += s.elapsed_time(e)
    del eng, model
    torch.cuda.empty_cache()
    return sorted(((k, c, ms) for k, (c, ms) in agg.items()), key=lambda r: -r[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--bf3", action="store_true")
    ap.add_argument("--shapes", type=int, default=99)
    ap.add_argument("--auto", type=int, default=0, help="search the N heaviest GEMM shapes of a profiled eager step (skipping --skip)")
    ap.add_argument("--kinds", default="", help="with --auto: only table kinds in this comma list (e.g. 2 = k=3 weight gradients)")
    ap.add_argument("--skip", type=int, default=0, help="with --auto: skip the first K shapes of the ranking (searched before)")
    ap.add_argument("--tiles", default="1,2,3,5,6", help="output tiles tried (0: 128x128, 1: 64x128, 2: 64x64, 3: 32x128, 4: 256x64, 5: 32x32, 6: 64x32)")
    ap.add_argument("--splits", default="4,8,16", help="wgrad token splits tried for the fused K=1 backward")
    ap.add_argument("--fwd", action="store_true", help="search the forward plans (kind 0, or 3 with --bf3) instead of the backward ones")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plan_search.json"))
    args = ap.parse_args()
    TILES[:] = [int(x) for x in args.tiles.split(",")]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model0, _ = build_model("deepfake_exp10.yaml")             # pristine CPU copy: TrainStep rebinds the parameters it is given

    def measure(rows):
        txt = ";".join("%d,%d,%d,%d,%d,%d,%d,%d" % (*k, a, b) for k, (a, b) in rows.items())
        check(lib.avdf_tuned_extra(ctypes.c_char_p(txt.encode()) if txt else None), "tuned_extra")
        model = copy.deepcopy(model0).to(dev)
        eng = TrainStep(model, args.batch, STREAMS, lr=1e-4, fwd_bf16x3=args.bf3, bwd_bf16x3=args.bf3)
        g = torch.Generator().manual_seed(1)
        for b in range(args.batch):
            for dst in eng.streams[b]:
                dst.copy_(torch.randn(dst.shape, generator=g))
            eng.ngt[b] = 0
        eng.gt.zero_()
        eng.gt[0, 0, 0], eng.gt[0, 0, 1], eng.ngt[0] = 100.0, 260.0, 1
        eng.capture()
        for _ in range(8):
            eng.step()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.step()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps)
        del eng, model
        torch.cuda.empty_cache()
        return args.batch / best

    todo = shapes(args.batch, args.bf3, args.fwd)[:args.shapes]
    if args.auto:
        ranked = ranked_shapes(model0, args.batch, args.bf3, dev)
        if args.kinds:
            ranked = [r for r in ranked if r[0][0] in [int(x) for x in args.kinds.split(",")]]
        for k, c, ms in ranked[:args.skip + args.auto]:
            print(f"  {ms:7.3f} ms/step  {k}", flush=True)
        todo = [(k, c) for k, c, _ in ranked[args.skip:args.skip + args.auto]]
    chosen, report = {}, []
    cur = measure(chosen)
    print(f"baseline (table in the tree): {cur:.1f} clips/s", flush=True)
    for key, cands in todo:
        res = {}
        for a, b in (cands(tuple(int(x) for x in args.splits.split(","))) if cands is pair_cands else cands()):
            trial = dict(chosen)
            trial[key] = (a, b)
            res[(a, b)] = measure(trial)
            print(f"  {key} plan ({a},{b}): {res[(a, b)]:.1f}", flush=True)
        (a, b), v = max(res.items(), key=lambda kv: kv[1])
        keep = v > cur * 1.003
        if keep:
            again = measure({**chosen, key: (a, b)})         # confirm once: single timings carry ~0.3 % noise
            keep = again > cur * 1.003
            v = min(v, again)
        if keep:
            chosen[key], cur = (a, b), v
        report.append({"shape": list(key), "best": [a, b], "clips_per_s": round(v, 1), "kept": keep,
                       "all": {f"{x},{y}": round(z, 1) for (x, y), z in res.items()}})
        print(f"{key}: best ({a},{b}) {v:.1f} -> {'kept' if keep else 'not kept'}; current {cur:.1f}", flush=True)
        json.dump({"batch": args.batch, "bf3": args.bf3, "chosen": {",".join(map(str, k)): list(v) for k, v in chosen.items()},
                   "final": round(cur, 1), "report": report}, open(args.out, "w"), indent=1)
    print("rows for gemm_tuned.inc:")
    for k, (a, b) in chosen.items():
        print("    {%d, %d, %d, %d, %d, %d, %d, %d}," % (*k, a, b))


if __name__ == "__main__":
    main()
