"""Data-parallel product path on the GPU box (one GPU, two ranks sharing it, gloo for the collectives): engine.TrainStep with
world size 2 must reproduce the single-process step on the union of the two shards, and engine.run_sharded_inference must
return what a single process returns.  RCCL itself needs one GPU per rank (the driver's 8-GPU run); the protocol -- broadcast
of the initial weights, all-reduce of the [#positives, #real clips] pair, bucketed asynchronous gradient all-reduces between
the backward phases, LR rule, loss reporting -- is backend independent and is what this file pins."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(video_input_dim=8, audio_input_dim=88, max_seq_len=192, embd_dim=32, fpn_dim=32, head_dim=32, n_head=4)
STREAMS = [(192, 8), (96, 64), (383, 24)]


def _build(seed, yaml_name="deepfake_exp10.yaml"):
    from audio_visual_deepfake_detection_amd import build_model
    torch.manual_seed(seed)
    model, cfg = build_model(yaml_name, **SMALL)
    model = model.to("cuda:0")
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    model.interpolator.dp1.p = 0.0
    return model


def _clip(i):
    g = torch.Generator().manual_seed(4000 + i)
    ss = [torch.randn(shp, generator=g) for shp in STREAMS]
    segs = None
    if i % 2 == 0:
        segs = torch.tensor([[20.0 + 7 * i, 60.0 + 5 * i], [100.0, 118.0]])
    return ss, segs


def _fill(eng, first):
    for b in range(eng.B):
        ss, segs = _clip(first + b)
        for dst, src in zip(eng.streams[b], ss):
            dst.copy_(src)
        eng.gt[b].zero_()
        eng.ngt[b] = 0 if segs is None else len(segs)
        if segs is not None:
            eng.gt[b, :len(segs)].copy_(segs)


def _train_worker(rank, world, port, out_path, yaml_name, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)          # before anything touches the GPU
    sys.path.insert(0, ROOT)
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    model = _build(100 + 13 * rank, yaml_name)                             # DIFFERENT weights per rank: rank 0's must win
    eng = TrainStep(model, 4, STREAMS, lr=1e-3 * world, weight_decay=0.05, clip_grad_l2norm=1.0, use_graph=True,
                    process_group=dist.group.WORLD, overlap=overlap)
    _fill(eng, rank * 4)
    eng.capture()
    for _ in range(2):
        eng.step()
    ls = eng.losses()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"losses": ls, "flat_p": eng.flat_p.cpu(), "exp_avg": eng.exp_avg.cpu(), "ema": eng.ema.cpu(),
                    "buckets": eng.bucket_ranges, "norm": float(model.loss_normalizer)}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("yaml_name,overlap", [("deepfake_exp10.yaml", True), ("deepfake_exp10.yaml", False),
                                               ("deepfake_exp5.yaml", True)])
def test_two_rank_train_step_equals_single_process_step(tmp_path, yaml_name, overlap):
    """TrainStep(world=2, 4 clips per rank) == TrainStep(world=1, the same 8 clips) after two steps: global losses, the
    loss normaliser, Adam's first moment (linear in the summed gradient) and the weights.  exp5 adds the L1 reconstruction
    term whose mean runs over the real clips of the GLOBAL batch."""
    import torch.multiprocessing as mp
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    out_path = str(tmp_path / "rank0.pt")
    port = 29500 + (os.getpid() * 7 + hash(yaml_name) + int(overlap)) % 2000
    mp.spawn(_train_worker, args=(2, port, out_path, yaml_name, overlap), nprocs=2, join=True)
    got = torch.load(out_path, weights_only=False)
    model = _build(100, yaml_name)
    eng = TrainStep(model, 8, STREAMS, lr=2e-3, weight_decay=0.05, clip_grad_l2norm=1.0, use_graph=True)
    _fill(eng, 0)
    eng.capture()
    for _ in range(2):
        eng.step()
    ref = eng.losses()
    torch.cuda.synchronize()
    assert [b[0] for b in got["buckets"]] == [b[0] for b in eng.bucket_ranges]
    assert len(got["buckets"]) == (2 if "exp5" in yaml_name else 3)
    for k, v in ref.items():
        assert abs(got["losses"][k] - v) < 1e-5 * max(1.0, abs(v)), (k, got["losses"][k], v)
    assert abs(got["norm"] - float(model.loss_normalizer)) < 1e-4
    m_ref, m_got = eng.exp_avg.cpu(), got["exp_avg"]
    # (exp5: the L1 term's gradient is a sign function of reco - x, whose near-zero elements flip with summation order)
    assert float((m_got - m_ref).abs().max()) < 1e-3 * float(m_ref.abs().max())
    # weights: Adam normalises every element's step to ~lr, so only elements with a gradient well above the fp32 summation
    # noise are comparable to 1e-5 of the weight scale
    p_ref, p_got = eng.flat_p.cpu(), got["flat_p"]
    sel = m_ref.abs() > 1e-3 * m_ref.abs().max()
    assert float((p_got - p_ref)[sel].abs().max()) < (1e-3 if "exp5" in yaml_name else 2e-4)
    assert float((got["ema"] - eng.ema.cpu())[sel].abs().max()) < 1e-6


def _infer_worker(rank, world, port, root, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from audio_visual_deepfake_detection_amd.engine import run_sharded_inference
    model, ds = _infer_setup(root)
    res = run_sharded_inference(model, ds, batch=2, process_group=dist.group.WORLD, output_folder=out_path if rank == 0 else None)
    assert (res is None) == (rank != 0)
    dist.barrier()
    dist.destroy_process_group()


def _infer_setup(root):
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.libs.datasets import make_inference_dataset
    torch.manual_seed(5)
    model, _ = build_model("deepfake_exp10.yaml", video_input_dim=8, audio_input_dim=24, embd_dim=32, fpn_dim=32, head_dim=32)
    model = model.to("cuda:0").eval()
    kw = dict(crop_ratio=None, default_fps=None, downsample_rate=0, video_feat_folder=os.path.join(root, "video"),
              audio_feat_folder=None, audio_byola_feat_folder=os.path.join(root, "byola"),
              audio_emo_feat_folder=os.path.join(root, "emo"), audio_file_ext=".npy", num_classes=1, input_dim=0,
              video_input_dim=8, audio_input_dim=24, feat_stride=1, num_frames=1,
              test_folder=os.path.join(root, "test_folder"), trunc_thresh=0.5, max_seq_len=768, force_upsampling=True)
    return model, make_inference_dataset("deepfake_video_audioEmoBYOLA_inference", False, ["test"], 1, **kw)


def test_sharded_inference_runner_equals_clip_by_clip_plugin_path(tmp_path):
    """BASELINE config 5's localisation leg: 7 clips over 2 ranks (shards of 4 and 3, last launch partially filled), gathered
    on rank 0 in dataset order == `model.eval()([item])` clip by clip in one process (the reference's inference loop)."""
    import json
    import torch.multiprocessing as mp
    from oracle.synth_data import make_synth_tree
    root = str(tmp_path)
    make_synth_tree(root, seed=9, n=7)
    out = str(tmp_path / "sub1")
    mp.spawn(_infer_worker, args=(2, 29500 + (os.getpid() * 3 + 11) % 2000, root, out), nprocs=2, join=True)
    got = json.load(open(os.path.join(out, "data_left.json")))
    model, ds = _infer_setup(root)
    assert [g["video_id"] for g in got] == [ds[i]["video_id"] for i in range(len(ds))]
    for i, g in enumerate(got):
        ref = model([ds[i]])[0]
        assert len(g["scores"]) == ref["segments"].shape[0]
        assert np.allclose(sorted(g["scores"]), sorted(ref["scores"].tolist()), atol=1e-5)
        a, b = np.asarray(g["segments"]).reshape(-1, 2), ref["segments"].numpy()
        assert np.allclose(a[np.lexsort(a.T)], b[np.lexsort(b.T)], atol=1e-3)
        assert abs(g["video_cls"][0] - float(ref["video_cls"][0])) < 1e-4


def test_bench_gpus_2_starts_its_own_ranks_and_reports_data_parallel():
    """The driver's command form, `python bench.py --gpus 2 ...` with NO rendezvous in the environment, on this one-GPU box:
    bench.py starts the two ranks itself (round-3 verdict: it used to run one process silently); with
    AVDF_DIST_BACKEND=gloo the ranks share the GPU and the collectives go through the host -- the measured
    configuration is one rank per GPU over RCCL, this pins the launch + the N-rank line (n_gpus, global batch,
    data_parallel object, losses from every rank)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(AVDF_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                        "--batch", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 4 and line["config"]["parallelism"] == "dp2"
    dpo = line["data_parallel"]
    assert dpo["grad_bytes"] > 100e6 and len(dpo["buckets_mb"]) >= 2 and dpo["ms_per_step_overlapped"] > 0
    assert np.isfinite(line["losses"]["final_loss"]) and line["value"] > 0
    assert "cpu_baseline" not in line                      # rank 0 at N = 1 only
