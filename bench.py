"""Headline benchmark: AV clips/s (768-frame) for one full training step (K1 interpolation ->
UMMAFormer forward -> losses -> backward -> clipped AdamW + EMA) of the exp10 AV-fusion model on
N MI355X GPUs of one node, one process per GPU, clip-level data parallel (weak scaling:
8 clips per GPU per step; gradient buckets all-reduced over RCCL while the next backward phase runs --
the line's `data_parallel` object reports the exposed all-reduce time and the step time without overlap).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant MFMA conv
kernel, per-launch HIP-event timing) and, at N=1, `cpu_baseline` (the CPU oracle -- a parity-pinned
restatement of the reference path -- timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA dense peak (= fp32 vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA dense peak (~2.5 PF)
STREAMS = [(768, 256), (383, 2048), (1535, 768)]      # LAV-DF video, BYOL-A, Emotion2Vec of a 768-frame clip


def synth_clip(i):
    """N(0,1) features from seed 1000+i; clip i fake iff i%2==0 with 1-3 segments of
    U(0.2,1.5) s at a uniform start, seed 2000+i; seconds -> feature grid: sec*fps/feat_stride - 0.5."""
    g = torch.Generator().manual_seed(1000 + i)
    feats = [torch.randn(shp, generator=g, dtype=torch.float32) for shp in STREAMS]
    segs = None
    if i % 2 == 0:
        rs = np.random.RandomState(2000 + i)
        k = int(rs.randint(1, 4))
        ln = rs.uniform(0.2, 1.5, size=k)
        st = rs.uniform(0.0, 30.72 - 1.5, size=k)
        segs = np.stack([st, st + ln], 1).astype(np.float32) * 25.0 - 0.5
    return feats, segs


def load_batch(eng, first_clip):
    for b in range(eng.B):
        feats, segs = synth_clip(first_clip + b)
        for dst, src in zip(eng.streams[b], feats):
            dst.copy_(src)
        if hasattr(eng, "gt"):
            eng.gt[b].zero_()
            n = 0 if segs is None else len(segs)
            if n:
                eng.gt[b, :n].copy_(torch.from_numpy(segs))
            eng.ngt[b] = n


def _profile_file(stem):
    """Newest committed PMC summary profiles/rNN_*_<stem>.json (collected by separate rocprofv3 --pmc passes, see
    tools/pmc_summary.py / tools/pmc_mfma.py) -- these are STATIC artefacts of the commit named inside them, not a
    measurement of this run; the bench line says so."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{stem}.json")))
    return files[-1] if files else None


def _norm_tag(tag):
    return tag.replace("fwd3", "fwd").replace("dgrad3", "dgrad").replace("bwdpair3", "bwdpair")


def pmc_traffic(tag):
    """HBM bytes per launch of the kernel class `tag` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
    correction of MI355X_MICROARCH.md + WRITE_SIZE, separate passes) -- or None."""
    path = _profile_file("pmc_traffic")
    try:
        d = json.load(open(path))
        ent = d["by_tag"][_norm_tag(tag)]
        return {"bytes_per_launch": ent["bytes_per_launch"], "fetch_bytes": ent["fetch_bytes"], "write_bytes": ent["write_bytes"],
                "static": True, "source": os.path.relpath(path, ROOT), "measured_at_commit": d.get("commit")}
    except Exception:
        return None


def pmc_mfma_util(tag):
    """MFMA-pipe utilisation of the kernel class `tag` (SQ_VALU_MFMA_BUSY_CYCLES per launch / (1024 SIMDs x un-instrumented
    launch duration x 2.4 GHz)) from the committed PMC pass -- or None."""
    path = _profile_file("pmc_mfma_util")
    try:
        d = json.load(open(path))
        return {"kernel": d["by_tag"][_norm_tag(tag)]["mfma_util"], "all_conv_kernels": d["all_conv_kernels"], "static": True,
                "source": os.path.relpath(path, ROOT), "measured_at_commit": d.get("commit")}
    except Exception:
        return None


def algorithmic_bytes(shape):
    """HBM bytes one launch of a conv / Linear GEMM has to move (fp32): every operand read once, every result written once.
    shape = ops._timed's record (kind, K, stride, B, Tin, Cin, Cout, problems); activation pre-images (ypre) not counted."""
    kind, K, stride, B, Tin, Cin, Cout, n = shape
    x, y, w = B * Tin * Cin, B * (Tin // stride) * Cout, Cout * Cin * K
    words = {"fwd": x + w + y, "dgrad": y + w + x, "wgrad": y + x + w, "pair": y + x + w + x + w}[kind]
    return 4 * n * words


def conv_flops_per_clip(cfg_T=768, Cin=3072, D=256):
    """Necessary forward MFMA work per clip (21.58 GF; x3 for fwd+bwd)."""
    return 21.58e9


def cpu_baseline(batch, budget_s=15.0):
    """CPU oracle (port of the reference path, pinned to it by tests/golden) timed on a bounded sample:
    repeated fwd+bwd training steps of `batch` exp10 clips until ~budget_s seconds elapsed."""
    from oracle import ummaformer_ref as R
    cfg = R.make_cfg()
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    sd = R.synth_state_dict(R.param_shapes(cfg), 0, trained_like=False)
    sd = {k: v.clone().requires_grad_(not k.startswith("interpolator.expansion")) for k, v in sd.items()}
    xs, gts = [], []
    for i in range(batch):
        feats, segs = synth_clip(i)
        xs.append(R.interp_concat([f.numpy() for f in feats], 768))
        gts.append(None if segs is None else torch.from_numpy(segs))
    x = torch.stack(xs)
    mask = torch.ones((batch, 1, 768))
    labels = [None if g is None else torch.zeros(len(g), dtype=torch.int64) for g in gts]
    done, t0 = 0, time.time()
    norm = cfg["init_loss_norm"]
    while True:
        out = R.model_forward(sd, cfg, x, mask, training=True)
        ls, norm, _ = R.losses(cfg, out, gts, labels, norm)
        ls["final_loss"].backward()
        for v in sd.values():
            v.grad = None
        done += 1
        el = time.time() - t0
        if el > budget_s or done >= 500:
            break
    return {"value": round(done * batch / el, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{done} fwd+bwd steps of {batch} exp10 clips (T=768, C_in=3072, fp32; K1 interpolation and the optimiser "
                      f"step are NOT in the CPU timing) in {el:.1f}s on {threads} threads"}


def cpu_baseline_infer(budget_s=15.0):
    """CPU oracle of the inference path (K1 -> forward -> decode -> NMS, one clip per call like inference.py) on a bounded
    sample: clips until ~budget_s seconds elapsed."""
    from oracle import ummaformer_ref as R
    cfg = R.make_cfg(input_dim=sum(c for _, c in STREAMS))
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    sd = R.synth_state_dict(R.param_shapes(cfg), 0, trained_like=False)
    done, t0 = 0, time.time()
    while True:
        feats, _ = synth_clip(done)
        x = R.interp_concat([f.numpy() for f in feats], 768, always=(False,) + (True,) * (len(feats) - 1)
                            if len(feats) > 1 else (True,))
        R.infer_clip(sd, cfg, x)
        done += 1
        el = time.time() - t0
        if el > budget_s or done >= 200:
            break
    return {"value": round(done / el, 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{done} clips (K1 interpolation + exp10 forward + decode + NMS, fp32) in {el:.1f}s on {threads} threads"}


def cpu_baseline_e2e(frames_per_sample=24):
    """CPU oracles of the raw-media path on a SHORT sample (C3D alone is 3.5 TFLOP per 768-frame clip: minutes on the host):
    one clip of `frames_per_sample` frames at 224x224 + the matching 16 kHz audio through the three encoder restatements,
    scaled linearly to 768 frames (every encoder is linear in clip length but the Emotion2Vec attention, which is
    quadratic -- the scaling flatters the CPU), plus one full-length localisation pass."""
    from oracle import byola_ref as BR, c3d_ref as CR, emotion2vec_ref as ER
    from audio_visual_deepfake_detection_amd.extractors import AudioNTT2020Task6, Emotion2Vec, C3DVideoEncoder
    threads = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(threads)
    torch.manual_seed(7)
    sd_b = AudioNTT2020Task6().eval().state_dict()
    sd_e = Emotion2Vec().state_dict()
    sd_c = C3DVideoEncoder().state_dict()
    g = torch.Generator().manual_seed(3000)
    n = frames_per_sample
    pcm = torch.rand(int(n / 25.0 * 16000), generator=g) - 0.5
    frames = torch.randint(0, 256, (n, 224, 224, 3), generator=g, dtype=torch.uint8)
    t0 = time.time()
    with torch.no_grad():
        CR.video_features(sd_c, frames, chunk=n + 8)                 # one chunk of n + 8 frames (the product's tail trim)
        BR.byola_forward(sd_b, BR.log_mel(pcm)[None, None])
        ER.forward(sd_e, pcm[None])
    t_enc = (time.time() - t0) * 768.0 / n
    loc = cpu_baseline_infer(budget_s=3.0)
    t_clip = t_enc + 1.0 / loc["value"]
    return {"value": round(1.0 / t_clip, 4), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"encoder restatements (C3D + log-mel/BYOL-A + Emotion2Vec, fp32) on one {n}-frame clip scaled x{768 // n} to "
                      f"768 frames ({t_enc:.1f}s per clip) + the localisation oracle ({1.0 / loc['value']:.2f}s per clip), {threads} threads"}


def self_launch(n):
    """`python bench.py --gpus N` with no rendezvous in the environment: start the N ranks ourselves (the single-command
    multi-device start of the reference: train.py:60-61,101 wraps the model in nn.DataParallel over cfg['devices']).
    One child `torch.distributed.run` with one worker per GPU over RCCL; this parent makes NO GPU call (counting devices
    does not initialise the runtime) and exits with the child's code.  Fewer than N visible devices is an error, not a
    silent single-GPU run -- unless AVDF_DIST_BACKEND=gloo asks for the rehearsal mode in which ranks share devices."""
    import socket
    import subprocess
    backend = os.environ.get("AVDF_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, found {ndev}; refusing to run a smaller job under that label "
              f"(AVDF_DIST_BACKEND=gloo rehearses the N-rank path on fewer devices)", file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def rehearse(args, world, rank):
    """AVDF_BENCH_REHEARSE=1: the launch + rendezvous + reporting protocol of an N-rank run without the device work
    (tests/test_bench_launch_cpu.py on a GPU-less host): every rank joins the group, proves the collective with a SUM
    all-reduce and the MAX-over-ranks timing reduction, rank 0 prints the line's launch-dependent fields."""
    backend = os.environ.get("AVDF_DIST_BACKEND", "gloo")
    assert backend == "gloo", "the rehearsal runs on the host"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.distributed.init_process_group("gloo")
    import importlib.util
    spec = importlib.util.spec_from_file_location("avdf_dp", os.path.join(ROOT, "audio_visual_deepfake_detection_amd", "dp.py"))
    dp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dp)                      # by path: the package import loads libavdf_hip.so, not needed here
    lo, hi = dp.shard_range(world * args.batch, rank, world)
    t = torch.tensor([float(hi - lo), float(rank)], dtype=torch.float64)
    torch.distributed.all_reduce(t)
    tmax = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    torch.distributed.barrier()
    if rank == 0:
        print(json.dumps({"rehearsal": True, "n_gpus": world, "global_batch": int(t[0].item()), "rank_sum": int(t[1].item()),
                          "max_over_ranks_s": float(tmax.item()), "config": {"parallelism": f"dp{world}"}}), flush=True)
    torch.distributed.destroy_process_group()
    return 0


def also_batch50(cfg, dev, steps=10, warmup=3):
    """The training step of the headline at 50 clips per GPU (fresh model + engine; fp32, hipGraph), timed like the main
    region.  Reported under `also`, never as `value`."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    torch.manual_seed(0)
    model, _ = build_model("deepfake_exp10.yaml")
    eng = TrainStep(model.to(dev), 50, STREAMS, lr=cfg["opt"]["learning_rate"], weight_decay=cfg["opt"]["weight_decay"],
                    clip_grad_l2norm=cfg["train_cfg"]["clip_grad_l2norm"])
    load_batch(eng, 0)
    eng.capture()
    for _ in range(warmup):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(50 * steps / dt, 2), "unit": "clips/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "warmup": warmup, "dtype": "f32", "note": "loader.batch_size of deepfake_exp10.yaml; informational, not the headline"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU per step")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the eager per-launch timing pass (tools/plan_ab.py)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the secondary measurement of the default line (`also`: the same step at the 50 clips per GPU of "
                         "deepfake_exp10.yaml's loader, 10 timed steps; never part of `value`)")
    ap.add_argument("--shape-table", default=None, metavar="PATH",
                    help="also write the per-(kernel class, GEMM shape) table of the roofline pass (launches, us, TFLOP/s, "
                         "time above an 80 %%-of-peak launch) to PATH: where the MFMA time of the step goes, shape by shape")
    ap.add_argument("--wgrad-side", action="store_true",
                    help="weight-gradient GEMMs on a second HIP stream (measured slower inside hipGraphs: 558 vs 611 clips/s)")
    ap.add_argument("--branch-streams", action="store_true", help="run the q/k/v chains on three HIP streams")
    ap.add_argument("--bf16x3-fwd", action="store_true",
                    help="training forward GEMMs with split-bf16 operands (807 vs 760 clips/s; losses agree to 1e-6, gradient "
                         "elements to ~5e-3 instead of 1e-4 -- not the parity-checked default)")
    ap.add_argument("--bf16x3-bwd", action="store_true",
                    help="training backward GEMMs (data gradients, K = 1 weight gradients) with split-bf16 operands; opt-in "
                         "like --bf16x3-fwd (tolerance stated in tests/test_gpu_engine.py)")
    ap.add_argument("--bf16x3", action="store_true",
                    help="infer / e2e modes: split-bf16 forward GEMMs (~5e-6 relative error; exact fp32 is the default)")
    ap.add_argument("--workload", default="av", choices=["av", "audio"],
                    help="av: exp10 AV fusion, C_in = 3072 (headline; BASELINE configs 3/4); audio: BYOL-A-only features, "
                         "C_in = 2048 (BASELINE config 2) through the same exp10 architecture")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "e2e"],
                    help="train (headline), infer (K1 -> fwd -> decode+NMS), e2e (raw PCM + frames -> 3 encoders -> infer)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
              f"(or run `python bench.py --gpus {args.gpus}` with no WORLD_SIZE set: it starts the ranks itself)", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("AVDF_BENCH_REHEARSE") == "1":
        sys.exit(rehearse(args, world, rank))
    # AVDF_DIST_BACKEND=gloo: rehearsal of the N>1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # collectives staged through the host); the measured configuration is always one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("AVDF_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1) if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=dev)
        else:
            torch.distributed.init_process_group(backend)
        pg = torch.distributed.group.WORLD

    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd import ops
    from audio_visual_deepfake_detection_amd.engine import TrainStep, InferStep

    ops.BRANCH_STREAMS = args.branch_streams
    ops.INFER_BF16X3 = bool(args.bf16x3)
    torch.manual_seed(rank)                    # per-rank init on purpose: TrainStep broadcasts rank 0's weights (dp.broadcast_params_)
    global STREAMS
    if args.workload == "audio":
        assert args.mode != "e2e", "--workload audio has no raw-media mode"
        STREAMS = [STREAMS[1]]
        model, cfg = build_model("deepfake_exp10.yaml", video_input_dim=0, audio_input_dim=2048)
    else:
        model, cfg = build_model("deepfake_exp10.yaml")
    model = model.to(dev)
    torch.manual_seed(1234 + rank)             # rank-offset drop-path / dropout streams

    B = args.batch
    extract = None
    if args.mode == "e2e":
        # BASELINE config 5 per-GPU share: raw 16 kHz PCM U(-0.5, 0.5) and 224x224 uint8 frames U{0..255} of 768-frame
        # clips -> BYOL-A, Emotion2Vec, C3D (random-init weights) -> K1 -> UMMAFormer -> NMS
        from audio_visual_deepfake_detection_amd.extractors import (LogMelFrontEnd, AudioNTT2020Task6, byola_features,
                                                                     Emotion2Vec, emotion2vec_features, C3DVideoEncoder,
                                                                     video_features, extract_clip_batch)
        torch.manual_seed(7)
        fe, byola = LogMelFrontEnd().to(dev), AudioNTT2020Task6().to(dev).eval()
        e2v, c3d = Emotion2Vec().to(dev), C3DVideoEncoder().to(dev)
        g = torch.Generator(device=dev).manual_seed(3000 + rank)
        pcm = torch.rand((B, 491520), generator=g, device=dev) - 0.5
        frames = torch.randint(0, 256, (B, 768, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)

        audio_stream = torch.cuda.Stream(device=dev) if os.environ.get("AVDF_E2E_STREAMS", "1") == "1" else None

        def extract(eng):
            # audio encoders on a second HIP stream beside the C3D video encoder (42.5 -> 50.2 clips/s)
            extract_clip_batch(frames, pcm, c3d, byola, fe, e2v, eng.streams, audio_stream)
    if args.mode == "train":
        eng = TrainStep(model, B, STREAMS, lr=cfg["opt"]["learning_rate"] * world,        # train.py:60 LR rule
                        weight_decay=cfg["opt"]["weight_decay"], clip_grad_l2norm=cfg["train_cfg"]["clip_grad_l2norm"],
                        use_graph=not args.no_graph, process_group=pg, wgrad_side=args.wgrad_side, fwd_bf16x3=args.bf16x3_fwd,
                        bwd_bf16x3=args.bf16x3_bwd)
    else:
        eng = InferStep(model, B, STREAMS, use_graph=not args.no_graph, bf16x3=args.bf16x3)
    from audio_visual_deepfake_detection_amd import dp
    first_clip, _ = dp.shard_range(world * B, rank, world)            # clip ids of this rank's shard of the global batch
    load_batch(eng, first_clip)
    eng.capture()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def one_step():
        if extract is not None:
            extract(eng)
        eng.step()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    clips_per_s = world * B * args.steps / dt

    # ---- data-parallel exchange: how much of it is exposed ---------------------------------------------
    comm = None
    if world > 1 and args.mode == "train":
        def timed_steps(n):
            barrier()
            t = time.perf_counter()
            for _ in range(n):
                eng.step()
            barrier()
            return (time.perf_counter() - t) / n * 1e3
        eng.comm_events = ()
        exposed = []
        for _ in range(5):
            eng.comm_events = ()
            eng.step()
            torch.cuda.synchronize()
            exposed.append(eng.comm_events[0].elapsed_time(eng.comm_events[1]))
        eng.comm_events = None
        eng.overlap = False
        serial_ms = timed_steps(max(5, args.steps // 2))
        eng.overlap = True
        comm = {"grad_bytes": int(eng.flat_g.numel() * 4),
                "buckets_mb": [round((hi - lo) * 4 / 1e6, 1) for _, lo, hi in eng.bucket_ranges],
                "allreduce_exposed_ms_per_step": round(float(np.median(exposed)), 3),
                "ms_per_step_overlapped": round(dt / args.steps * 1e3, 3), "ms_per_step_not_overlapped": round(serial_ms, 3)}

    # ---- roofline pass: the same step, eager, HIP events around every MFMA conv launch -----------
    # A GPU-side spin is queued first so that the host has enqueued the whole step before the first kernel runs:
    # the events then bracket back-to-back GPU execution (no host launch gaps inside the brackets).
    roof = None
    if args.no_roofline:
        if rank == 0:
            print(json.dumps({"value": round(clips_per_s, 2), "ms_per_step": round(dt / args.steps * 1e3, 3)}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    bf3 = (args.mode != "train" and args.bf16x3) or (args.mode == "train" and args.bf16x3_bwd)   # arithmetic of the dominant class
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if bf3 else PEAK_F32_MFMA_TFLOPS
    if args.mode == "train":
        use_graph = eng.use_graph
        graphs, eng.graphs, eng.use_graph = eng.graphs, None, False
        run_eager = eng.step
    else:
        graphs, eng.graph = eng.graph, None
        run_eager = one_step
    run_eager()                      # eager warm-up
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(1000000)
    e1.record()
    torch.cuda.synchronize()
    spin = int(60.0 / max(e0.elapsed_time(e1), 1e-3) * 1000000)       # ~60 ms worth of spin cycles
    ops.profile_begin()
    nprof = 3 if args.mode != "e2e" else 1
    for _ in range(nprof):
        torch.cuda._sleep(spin)
        run_eager()
        torch.cuda.synchronize()
    recs = ops.profile_end()
    if args.mode == "train":
        eng.graphs, eng.use_graph = graphs, use_graph
    else:
        eng.graph = graphs
    agg = {}
    for tag, fl, s, e, shape in recs:
        a = agg.setdefault(tag, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += s.elapsed_time(e) * 1e-3
        a[2] += fl
        a[3] += algorithmic_bytes(shape) if shape is not None else 0.0
    if args.shape_table and rank == 0:
        by_shape = {}
        for tag, fl, s, e, shape in recs:
            a = by_shape.setdefault((tag, shape), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e) * 1e-3
            a[2] += fl
        rows = sorted(by_shape.items(), key=lambda kv: -(kv[1][1] - kv[1][2] / (0.8 * peak * 1e12)))
        with open(args.shape_table, "w") as f:
            f.write("# kernel class | (kind, K, stride, B, T, Cin, Cout, problems) | launches/step | us/launch | TFLOP/s | "
                    "us/step above an 80%-of-peak launch\n")
            for (tag, shape), (n, t, fl) in rows:
                f.write(f"{tag:24s} {str(shape):44s} {n / nprof:6.1f} {t / n * 1e6:9.1f} {fl / t / 1e12:8.1f} "
                        f"{(t - fl / (0.8 * peak * 1e12)) / nprof * 1e6:9.1f}\n")
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    tot_t = sum(v[1] for v in agg.values())
    tot_f = sum(v[2] for v in agg.values())
    tag, (cnt, tsec, fl, abytes) = dom
    ach = fl / tsec / 1e12
    roof = {"bound": "mfma", "kernel": tag, "achieved": round(ach, 2), "peak": round(peak, 1),
            "peak_note": ("bf16 dense MFMA peak / 3 (three bf16 MFMAs per product in bf16x3 mode)" if bf3 else
                          "f32-input MFMA dense peak (the arithmetic the path computes in)"),
            "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "traffic": pmc_traffic(tag) if args.mode == "train" else None,
            "algorithmic_bytes_per_launch": round(abytes / cnt),
            "mfma_util": pmc_mfma_util(tag) if args.mode == "train" else None,
            **({"frac_of_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4)} if bf3 else {}),
            "launches_per_step": cnt // nprof, "avg_launch_us": round(tsec / cnt * 1e6, 2),
            "flops_per_launch": round(fl / cnt),
            "all_mfma_kernels": {"achieved": round(tot_f / tot_t / 1e12, 2), "frac": round(tot_f / tot_t / 1e12 / peak, 4),
                                 "ms_per_step": round(tot_t / nprof * 1e3, 3)},
            "per_kernel": {k: {"n": v[0] // nprof, "ms_per_step": round(v[1] / nprof * 1e3, 3),
                               "tflops": round(v[2] / v[1] / 1e12, 2)} for k, v in sorted(agg.items())}}

    if roof.get("traffic"):
        roof["traffic"]["over_algorithmic"] = round(roof["traffic"]["bytes_per_launch"] / max(roof["algorithmic_bytes_per_launch"], 1), 2)
    losses = eng.losses() if args.mode == "train" else None          # a collective under DP: every rank calls it
    if rank == 0:
        line = {
            "metric": {"train": "AV clips/sec (768-frame) fwd+bwd", "infer": "AV clips/sec (768-frame) inference",
                       "e2e": "AV clips/sec (768-frame) raw media -> features -> localisation"}[args.mode],
            "value": round(clips_per_s, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "exp10 UMMAFormer training step (K1 interp -> fwd -> losses -> bwd -> "
                                   "clip+AdamW+EMA), T=768, C_in=%d, random-init weights" % sum(c for _, c in STREAMS)
                                   if args.mode == "train" else
                                   ("exp10 AV-fusion UMMAFormer inference (K1 interp -> fwd -> decode+NMS)" if args.mode == "infer"
                                    else "raw PCM (491520 samples) + 768x224x224 u8 frames -> log-mel/BYOL-A + Emotion2Vec + C3D "
                                         "-> K1 -> exp10 UMMAFormer inference -> decode+NMS, random-init weights"),
                       "features": ("LAV-DF 256 + BYOL-A 2048 + Emotion2Vec 768" if args.workload == "av"
                                    else "BYOL-A 2048 only (BASELINE config 2 shapes: C_in = 2048 through the exp10 architecture)"),
                       "clips_per_gpu": B, "global_batch": world * B, "seq_len": 768,
                       "parallelism": f"dp{world}", "hip_graph": bool(eng.use_graph),
                       "fwd_gemm": ("bf16x3" if ((args.mode != "train" and args.bf16x3) or args.bf16x3_fwd) else "f32"),
                       "bwd_gemm": ("bf16x3 (dgrad, K=1 wgrad)" if args.bf16x3_bwd else "f32"),
                       "dead_params_skipped": len(getattr(eng, "dead", []))},
        }
        if roof is not None:
            line["roofline"] = roof
            if args.mode == "train":
                fwd_bwd = 3 * conv_flops_per_clip() * clips_per_s / world / 1e12
                line["step_mfma_frac"] = round(fwd_bwd / PEAK_F32_MFMA_TFLOPS, 4)
        if comm is not None:
            line["data_parallel"] = comm
        if args.mode == "train":
            line["losses"] = losses
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = (cpu_baseline(2) if args.mode == "train" else
                                    cpu_baseline_infer() if args.mode == "infer" else cpu_baseline_e2e())
        if (world == 1 and args.mode == "train" and B == 8 and args.workload == "av" and not args.no_also
                and not (args.bf16x3_fwd or args.bf16x3_bwd or args.no_graph)):
            # secondary, informational: the same captured step at the batch size of the reference's own config
            # (deepfake_exp10.yaml: loader.batch_size 50) -- the 8-clip headline is config 4's per-GPU share, a latency regime
            try:
                line["also"] = {"train_50_clips_per_gpu": also_batch50(cfg, dev)}
            except Exception as exc:                                  # never lose the headline line over the extra
                line["also"] = {"error": repr(exc)[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
