"""Whole-path parity on a real MI355X against (a) the golden vectors produced by the REAL reference
(tests/golden/*.npz, see oracle/make_golden.py) and (b) the CPU oracle on fresh seeded inputs.

Tolerances are the north star's: 1e-3 on logits / offsets / IoU-bearing quantities (observed ~1e-5),
bit-exact kept point indices (compared as sets: near-tied fp32 scores may swap ranks, the reference's
own sorts are unstable)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(input_dim=96, max_seq_len=192, embd_dim=32, n_head=4, fpn_dim=32, head_dim=32)
MID = dict(input_dim=192, max_seq_len=384, embd_dim=128, n_head=4, fpn_dim=128, head_dim=128)
TOL = 1e-3


YAML = {"norecon": "deepfake_exp10.yaml", "the": "deepfake_exp13.yaml", "recon": "deepfake_exp5.yaml",
        "plain": "deepfake_exp10.yaml", "plainconv": "deepfake_exp10.yaml"}


def build(cfg_over, wseed, trained_like=True):
    from audio_visual_deepfake_detection_amd import build_model
    from oracle import ummaformer_ref as R
    cfg = R.make_cfg(**cfg_over)
    extra = {}
    if cfg["variant"] in ("plain", "plainconv"):
        extra = dict(model_name="LocPointTransformer", backbone_type="conv" if cfg["variant"] == "plainconv" else "convTransformer")
    vdim = cfg.get("video_dim", 8)
    model, _ = build_model(YAML[cfg["variant"]], video_input_dim=vdim, audio_input_dim=cfg["input_dim"] - vdim,
                           max_seq_len=cfg["max_seq_len"], embd_dim=cfg["embd_dim"], fpn_dim=cfg["fpn_dim"],
                           head_dim=cfg["head_dim"], n_head=cfg["n_head"], use_rel_pe=bool(cfg.get("use_rel_pe", False)), **extra)
    sd = R.synth_state_dict(R.param_shapes(cfg), wseed, trained_like)
    model.load_state_dict(sd, strict=True)
    return model.to("cuda:0"), sd, cfg


def synth_inputs(cfg, lens, seed):
    rs = np.random.RandomState(seed)
    B, C, T = len(lens), cfg["input_dim"], cfg["max_seq_len"]
    x = np.zeros((B, C, T), dtype=np.float32)
    for b, n in enumerate(lens):
        x[b, :, :n] = rs.standard_normal((C, n)).astype(np.float32)
    return torch.from_numpy(x)


def clips_from(x, lens, segments=None, frame_labels=None):
    out = []
    for b, n in enumerate(lens):
        d = dict(video_id=f"c{b}.mp4", feats=x[b, :, :n].clone(), fps=25.0, duration=n / 25.0, feat_stride=1.0,
                 feat_num_frames=1.0)
        if segments is not None:
            seg = segments[b]
            d["segments"] = None if seg is None else torch.as_tensor(seg, dtype=torch.float32)
            d["labels"] = None if seg is None else torch.zeros(len(seg), dtype=torch.int64)
            d["gt_frame_labels"] = None
            if frame_labels is not None and seg is not None:
                d["gt_frame_labels"] = torch.as_tensor(np.asarray(frame_labels[b], np.float32))
        out.append(d)
    return out


def maxerr(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


THE_SMALL = dict(variant="the", **SMALL)
RECON_SMALL = dict(variant="recon", **SMALL)
PLAIN_SMALL = dict(variant="plain", **SMALL)          # LocPointTransformer + convTransformer backbone
CONV_SMALL = dict(variant="plainconv", **SMALL)       # LocPointTransformer + conv backbone
RELPE_SMALL = dict(use_rel_pe=True, **SMALL)          # relative position bias in every local attention
AUDIO_EXP10 = dict(input_dim=2048, video_dim=0)                  # BASELINE config 2: BYOL-A features only, full-size exp10 architecture
EXP5 = dict(variant="recon", input_dim=1024, video_dim=256)      # BASELINE config 1: deepfake_exp5.yaml at full size


@pytest.mark.parametrize("name,over", [("fwd_small", SMALL), ("fwd_small_init", SMALL), ("fwd_mid", MID),
                                       ("fwd_exp10", {}), ("fwd_audio_exp10", AUDIO_EXP10), ("fwd_exp5", EXP5),
                                       ("fwd_the_small", THE_SMALL),
                                       ("fwd_recon_small", RECON_SMALL), ("fwd_relpe_small", RELPE_SMALL),
                                       ("fwd_plain_small", PLAIN_SMALL),
                                       ("fwd_conv_small", CONV_SMALL)])
def test_forward_and_detections_match_reference_goldens(golden_dir, name, over):
    g = np.load(os.path.join(golden_dir, f"{name}.npz"), allow_pickle=True)
    model, sd, cfg = build(over, int(g["wseed"]), bool(int(g["trained_like"])))
    model.eval()
    model.allow_batched_eval = True
    lens = [int(v) for v in g["lens"]]
    x = synth_inputs(cfg, lens, int(g["xseed"]))
    clips = clips_from(x, lens)
    with torch.no_grad():
        xt, mask = model.preprocessing(clips)
        logits, offsets, pmask, cls_scores, _ = model.forward_features(xt, mask)
    e_log = maxerr(logits.cpu(), g["logits"][:, :, 0])
    e_off = maxerr(offsets.cpu(), g["offsets"])
    e_cls = maxerr(cls_scores.cpu(), g["cls_scores"])
    print(f"[{name}] logits {e_log:.3e} offsets {e_off:.3e} video-cls {e_cls:.3e}")
    assert e_log < TOL and e_off < TOL and e_cls < TOL
    # full eval path through the plugin API, one clip per call like inference.py
    model.allow_batched_eval = False
    for b, clip in enumerate(clips):
        res = model([clip])[0]
        k = int(g["n_segments"][b])
        assert res["segments"].shape[0] == k, (res["segments"].shape, k)
        ref_pts = g["kept_points"][b][:k]
        assert sorted(res["kept_points"].tolist()) == sorted(ref_pts.tolist()), f"{name} clip {b}: kept point sets differ"
        # match rows through the (bit-exact) point index, then compare seconds / scores
        order = {int(p): i for i, p in enumerate(res["kept_points"].tolist())}
        perm = torch.tensor([order[int(p)] for p in ref_pts], dtype=torch.int64)
        e_seg = maxerr(res["segments"][perm], g["segments"][b][:k]) if k else 0.0
        e_sc = maxerr(res["scores"][perm], g["scores"][b][:k]) if k else 0.0
        moved = int((perm != torch.arange(k)).sum())
        print(f"[{name}] clip {b}: {k} segments, seg err {e_seg:.3e} s, score err {e_sc:.3e}, rank moves {moved}")
        assert e_seg < TOL and e_sc < 1e-4
        if "video_cls" in res:
            assert maxerr(res["video_cls"], g["video_cls"][b]) < TOL
        assert res["labels"].dtype == torch.int64 and res["scores"].dtype == torch.float32


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("name,over", [("train_small", SMALL), ("train_mid", MID), ("train_the_small", THE_SMALL),
                                       ("train_recon_small", RECON_SMALL), ("train_relpe_small", RELPE_SMALL),
                                       ("train_plain_small", PLAIN_SMALL),
                                       ("train_conv_small", CONV_SMALL),
                                       ("train_exp10", {}),         # {} = full-size exp10
                                       ("train_audio_exp10", AUDIO_EXP10), ("train_exp5", EXP5)])
def test_training_losses_and_gradients_match_reference_goldens(golden_dir, name, over, accumulate):
    """accumulate=False: deterministic slab reductions, autograd owns the gradients.
    accumulate=True: the engine path -- every backward kernel adds straight into `param.main_grad`
    (float atomics / fused reductions) and autograd sees None for parameter gradients."""
    g = np.load(os.path.join(golden_dir, f"{name}.npz"), allow_pickle=True)
    model, sd, cfg = build(over, int(g["wseed"]))
    model.train()
    if accumulate:
        for p in model.parameters():           # same wiring as engine.TrainStep: .grad IS the accumulation buffer
            p.main_grad = torch.zeros_like(p)
            p.grad = p.main_grad
    for mod in model.modules():                 # same switches as the golden run (drop-path / dropout off)
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    if hasattr(model, "interpolator"):
        model.interpolator.dp1.p = 0.0
    lens = [int(v) for v in g["lens"]]
    x = synth_inputs(cfg, lens, int(g["xseed"]))
    segs = [np.asarray(s, np.float32) if f else None for s, f in zip(g["gt_segments"], g["is_fake"])]
    losses = model(clips_from(x, lens, segs, g["gt_frame_labels"] if cfg["variant"] == "the" else None))
    losses["final_loss"].backward()
    if cfg["variant"] == "the":
        assert abs(float(losses["reco_seg_loss"]) - float(g["reco_seg_loss"])) < 1e-5
    if cfg["variant"] == "recon":
        assert abs(float(losses["reco_loss"]) - float(g["reco_loss"])) < 1e-5
    got = np.array([float(losses[k].detach()) if k in losses else 0.0 for k in ("cls_loss", "reg_loss", "reco_cls_loss", "final_loss")])
    print(f"[{name}] losses got {got} ref {g['losses']}")
    assert np.allclose(got, g["losses"], rtol=1e-4, atol=1e-5)
    assert abs(model.loss_normalizer - float(g["loss_normalizer"])) < 1e-3
    named = dict(model.named_parameters())
    worst = 0.0
    f64_path = os.path.join(golden_dir, f"{name}_f64.npz")
    f64 = np.load(f64_path, allow_pickle=True) if os.path.exists(f64_path) else None       # full-size fixtures only
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        p = named[key[6:]]
        flat = (p.main_grad if accumulate else p.grad).detach().flatten().cpu()
        stride = max(1, flat.numel() // 4096)
        sample = flat[::stride][:4096].double()
        ref = torch.from_numpy(g[key]).double()
        rel = float((sample - ref).abs().max() / (ref.abs().max() + 1e-12))
        ssum = g["gsum::" + key[6:]]
        rel_sum = abs(float(flat.double().abs().sum()) - ssum[1]) / (ssum[1] + 1e-12)
        # element tolerance: 1e-3 of the tensor's scale -- widened (a) to 3x the REFERENCE's own fp32 error against float64
        # ground truth where that is larger (tests/golden/<name>_f64.npz: exp5's L1 / ReLU kinks put the reference itself
        # up to 1e-2 away from the truth on some tensors), and (b) to 2e-2 on the DeepInterpolator convs, see below
        tol_el = TOL
        if f64 is not None and ("referr::" + key[6:]) in f64.files:
            tol_el = max(TOL, 3.0 * float(f64["referr::" + key[6:]]))
        if f64 is not None and "contraction.down_" in key:
            tol_el = max(tol_el, 2e-2)
        worst = max(worst, rel if tol_el == TOL else 0.0, rel_sum)
        # full-size models: every DownBlock / UpBlock is conv -> InstanceNorm -> LeakyReLU(0.2), whose derivative jumps
        # between 0.2 and 1 at zero.  Among the ~10^6 InstanceNorm outputs of a batch a handful lie within fp32 rounding
        # distance of zero (|h| ~ 1e-6); there the sign -- hence the gradient of that (clip, channel) row -- depends on the
        # summation order of the conv in front.  Measured against float64 ground truth (test_interpolator_gradient_
        # differences_are_leaky_relu_flips below; tests/golden/train_exp10_f64.npz): ONE flipped element in down_4 accounts
        # for the whole difference, every other row agrees to 2e-6.  The element sample of the affected tensors is therefore
        # held to 2e-2, their L1 norm to the normal tolerance.
        assert rel < tol_el and rel_sum < TOL, (key, rel, rel_sum)
    print(f"[{name}] worst relative gradient error {worst:.3e}")
    if accumulate:
        assert all(p.grad is p.main_grad for p in named.values()), "accumulate path must not replace .grad"
        # a per-level Scale() whose level has no positive point has an exactly-zero (not missing) gradient
        dead = sorted(k for k, p in named.items() if float(p.main_grad.abs().sum()) == 0.0
                      and not k.startswith("reg_head.scale."))
    else:
        dead = sorted(k for k, p in named.items() if p.grad is None)
    assert len(dead) == int(g["num_dead"]), (len(dead), int(g["num_dead"]))


def test_fresh_inputs_against_cpu_oracle():
    """Seeded inputs that no fixture covers, incl. a 1-frame clip and a fully valid clip."""
    from oracle import ummaformer_ref as R
    model, sd, cfg = build(SMALL, 11)
    model.eval()
    lens = [192, 1, 96, 37]
    x = synth_inputs(cfg, lens, 12)
    for b, clip in enumerate(clips_from(x, lens)):
        res = model([clip])[0]
        o = R.infer_clip(sd, cfg, x[b, :, :lens[b]], dict(fps=25.0, duration=lens[b] / 25.0, feat_stride=1.0,
                                                          feat_num_frames=1.0))
        assert res["segments"].shape[0] == o["segments"].shape[0]
        assert sorted(res["kept_points"].tolist()) == sorted(o["kept_points"].tolist())
        k = res["segments"].shape[0]
        if k:
            order = {int(p): i for i, p in enumerate(res["kept_points"].tolist())}
            perm = torch.tensor([order[int(p)] for p in o["kept_points"].tolist()])
            assert maxerr(res["segments"][perm], o["segments"]) < TOL
            assert maxerr(res["scores"][perm], o["scores"]) < 1e-4


def test_batched_step_with_droppath_runs_and_is_finite():
    model, sd, cfg = build(SMALL, 13)
    model.train()
    lens = [192, 180, 175, 192]
    x = synth_inputs(cfg, lens, 14)
    segs = [np.array([[10.0, 30.0]], np.float32), None, np.array([[5.0, 9.0], [100.0, 140.0]], np.float32), None]
    losses = model(clips_from(x, lens, segs))
    losses["final_loss"].backward()
    assert all(torch.isfinite(v).all() for v in losses.values())
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n


def test_cpu_tensors_fail_loudly():
    from audio_visual_deepfake_detection_amd import ops
    with pytest.raises(RuntimeError):
        ops.layernorm(torch.zeros(2, 4, 8), torch.ones(1, 8, 1), torch.zeros(1, 8, 1))


def test_dataset_items_through_model(tmp_path):
    """Items of this package's dataset plugin (raw streams + crop window) through the model: the on-device K1
    interpolation + crop + padding must equal the CPU interpolation of the reference's dataset, and the
    training / eval entry points must accept the items unchanged."""
    import random
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.libs.datasets import make_dataset, make_data_loader
    from oracle import ummaformer_ref as R
    from oracle.synth_data import make_synth_tree, dataset_kwargs
    root = str(tmp_path)
    make_synth_tree(root, seed=3, n=4)
    ds = make_dataset("deepfake_video_audioEmoBYOLA", True, ["train"], **dataset_kwargs(root))
    model, _ = build_model("deepfake_exp10.yaml", video_input_dim=8, audio_input_dim=24, embd_dim=32, fpn_dim=32,
                           head_dim=32)
    model = model.to("cuda:0").train()
    random.seed(5)
    batch = [ds[i] for i in range(4)]
    x, mask = model.preprocessing(batch)
    for b, it in enumerate(batch):
        ref = R.interp_concat(it["streams"], 768)
        st, ed = it["crop"] if it["crop"] is not None else (0, 768)
        ref = ref[:, st:ed].t()
        n = ref.shape[0]
        assert torch.equal(x[b, :n].cpu(), ref), f"clip {b}: device K1+crop differs from the CPU interpolation"
        assert float(x[b, n:].abs().sum()) == 0.0 and float(mask[b].sum()) == n
    losses = model(batch)
    losses["final_loss"].backward()
    assert torch.isfinite(losses["final_loss"]).item()
    model.eval()
    res = model([batch[0]])[0]
    assert res["segments"].shape[1] == 2 and res["video_id"] == batch[0]["video_id"]


def test_device_prefetcher_feeds_identical_batches(tmp_path):
    """(f)-2: pinned, asynchronous H2D staging one batch ahead -- same tensors as the synchronous path, in order."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.libs.datasets import make_dataset, make_data_loader
    from audio_visual_deepfake_detection_amd.libs.datasets.prefetch import DevicePrefetcher
    from oracle.synth_data import make_synth_tree, dataset_kwargs
    root = str(tmp_path)
    make_synth_tree(root, seed=4, n=6)
    ds = make_dataset("deepfake_video_audioEmoBYOLA", False, ["train"], **dataset_kwargs(root))
    loader = make_data_loader(ds, False, None, batch_size=2, num_workers=0)
    model, _ = build_model("deepfake_exp10.yaml", video_input_dim=8, audio_input_dim=24, embd_dim=32, fpn_dim=32,
                           head_dim=32)
    model = model.to("cuda:0").train()
    plain = [model.preprocessing(b) for b in loader]
    n = 0
    for (x_ref, m_ref), batch in zip(plain, DevicePrefetcher(loader, "cuda:0", depth=2)):
        assert all(t.is_cuda for it in batch for t in it["streams"])
        x, m = model.preprocessing(batch)
        assert torch.equal(x, x_ref) and torch.equal(m, m_ref)
        n += 1
    assert n == len(plain) == 3


def test_av_loc_point_transformer_is_the_plain_model_on_concatenated_input():
    """`AVLocPointTransformer` (av_meta_arch.py:162) == `LocPointTransformer` with input_dim + audio_input_dim channels."""
    from audio_visual_deepfake_detection_amd import build_model
    kw = dict(max_seq_len=192, embd_dim=32, fpn_dim=32, head_dim=32, n_head=4, backbone_type="convTransformer")
    a, _ = build_model("deepfake_exp10.yaml", model_name="AVLocPointTransformer", video_input_dim=8, audio_input_dim=88, **kw)
    b, _ = build_model("deepfake_exp10.yaml", model_name="LocPointTransformer", video_input_dim=96, audio_input_dim=0, **kw)
    assert [k for k in a.state_dict()] == [k for k in b.state_dict()]
    b.load_state_dict(a.state_dict())
    a, b = a.to("cuda:0").eval(), b.to("cuda:0").eval()
    clip = dict(video_id="v", feats=torch.randn(96, 150), fps=25.0, duration=6.0, feat_stride=1.0, feat_num_frames=1.0)
    ra, rb = a([clip])[0], b([clip])[0]
    assert torch.equal(ra["segments"], rb["segments"]) and torch.equal(ra["scores"], rb["scores"]) and "video_cls" not in ra


def test_tinydataset_metadata_through_dataset_and_exp5_model(tmp_path, golden_dir):
    """BASELINE config 1: the reference's 12 tinydataset metadata files (tests/golden/tinydataset_meta.json) -> this package's
    `deepfake_video_audio` dataset (raw streams; K1 on the device) -> full-size exp5 model, one clip per call like inference.py;
    expected values come from the reference's own dataset class + model on the same seeded features (oracle/make_golden.py)."""
    from audio_visual_deepfake_detection_amd.libs.datasets import make_dataset
    from oracle.synth_data import tiny_tree_from_fixture, tiny_dataset_kwargs
    g = np.load(os.path.join(golden_dir, "tinydataset.npz"), allow_pickle=True)
    root = str(tmp_path)
    tiny_tree_from_fixture(root, os.path.join(golden_dir, "tinydataset_meta.json"), seed=500)
    ds = make_dataset("deepfake_video_audio", False, ["train"], **tiny_dataset_kwargs(root))
    assert len(ds) == int(g["n"]) == 12
    model, sd, cfg = build(EXP5, int(g["wseed"]))
    model.eval()
    for i in range(len(ds)):
        item = ds[i]
        x, mask = model.preprocessing([item])
        assert float(mask.sum()) == 768
        feats = x[0].t().cpu()                                       # [C, T] like the reference's `feats`
        assert np.array_equal(feats[0].numpy(), g[f"e{i}_row0"]) and np.array_equal(feats[-1].numpy(), g[f"e{i}_rowN"])
        assert np.allclose([float(feats.double().sum()), float(feats.double().abs().sum())], g[f"e{i}_sum"], rtol=1e-9)
        res = model([item])[0]
        k = int(g["n_segments"][i])
        assert res["segments"].shape[0] == k
        ref_pts = g["kept_points"][i][:k]
        assert sorted(res["kept_points"].tolist()) == sorted(ref_pts.tolist()), f"clip {i}: kept point sets differ"
        order = {int(p): j for j, p in enumerate(res["kept_points"].tolist())}
        perm = torch.tensor([order[int(p)] for p in ref_pts], dtype=torch.int64)
        assert maxerr(res["segments"][perm], g["segments"][i][:k]) < TOL
        assert maxerr(res["scores"][perm], g["scores"][i][:k]) < 1e-4
        assert maxerr(res["video_cls"], g["video_cls"][i]) < TOL


def test_interpolator_gradient_differences_are_leaky_relu_flips():
    """Why single gradient elements of the DeepInterpolator differ from the reference by up to ~2e-2 of the tensor's scale while
    every loss, logit and L1 norm agrees to 1e-5: float64 ground truth (the oracle, which is pinned to the reference in fp32,
    re-run in float64 on the CPU) shows that the HIP gradient w.r.t. each conv output is exact to ~2e-6 on every (clip, channel)
    row EXCEPT the rows in which one InstanceNorm output lies within fp32 rounding distance of zero and the HIP conv's summation
    order puts it on the other side: LeakyReLU's derivative there is 1 instead of 0.2 (or vice versa).  The test asserts that
    mechanism, row by row, on the full-size exp10 contraction with the weights / inputs of train_exp10."""
    import torch.nn.functional as F
    from audio_visual_deepfake_detection_amd import ops
    from oracle import ummaformer_ref as R
    torch.set_num_threads(16)
    model, sd, cfg = build({}, 21)
    model.train()
    lens = [768, 700, 768, 650]
    x = synth_inputs(cfg, lens, 23)
    mask = (torch.arange(768)[None, :] < torch.tensor(lens)[:, None]).float()[:, None, :]
    wv = torch.tensor([1.0, -0.7, 0.4, 1.3])
    p = "interpolator."

    def run_cpu(dtype):
        sdd = {k: v.to(dtype) for k, v in sd.items() if k.startswith(p)}
        pres, hs = [], []
        f, m = x.to(dtype).requires_grad_(True), mask.to(dtype)
        for i in range(5):
            y, m = R.masked_conv1d(f, m, sdd[f"{p}contraction.down_{i + 1}.conv_block.conv.weight"],
                                   sdd[f"{p}contraction.down_{i + 1}.conv_block.conv.bias"], 2)
            y.retain_grad()
            h = F.instance_norm(y, eps=1e-5)
            f = F.leaky_relu(h, 0.2)
            pres.append(y)
            hs.append(h.detach())
        c = F.leaky_relu(F.instance_norm(F.conv1d(f, sdd[p + "conv0.0.weight"]), eps=1e-5), 0.2)
        pooled = torch.cat((c.max(dim=2).values, c.mean(dim=2)), dim=1)
        hcl = F.relu(R.layer_norm_ct(F.linear(pooled, sdd[p + "conv1.weight"]).unsqueeze(-1), sdd[p + "bn1.weight"],
                                     sdd[p + "bn1.bias"])).squeeze(-1)
        cls = F.linear(hcl, sdd[p + "conv2.weight"], sdd[p + "conv2.bias"])
        (cls[:, 0] * wv.to(dtype)).sum().backward()
        return [t.grad for t in pres], hs, [t.detach() for t in pres]

    g64, h64, y64 = run_cpu(torch.float64)
    model.interpolator.dp1.p = 0.0
    f, m = x.transpose(1, 2).contiguous().to("cuda:0").requires_grad_(True), mask[:, 0].contiguous().to("cuda:0")
    pres = []
    ct = model.interpolator.contraction
    for blk in (ct.down_1, ct.down_2, ct.down_3, ct.down_4, ct.down_5):
        y, m = blk.conv_block(f, m, exact=True)
        y.retain_grad()
        f = ops.instnorm_lrelu(y, 1e-5, 0.2)
        pres.append(y)
    cls = model.interpolator.classifier(f)
    (cls[:, 0] * wv.to("cuda:0")).sum().backward()
    total_flips = 0
    for i in range(4, -1, -1):
        yh = pres[i].detach().transpose(1, 2).cpu().double()
        gh = pres[i].grad.transpose(1, 2).cpu().double()
        conv_err = float((yh - y64[i]).abs().max())
        assert conv_err < 3e-6 * float(y64[i].abs().max())                    # the forward conv itself is fp32-exact
        flips = (F.instance_norm(yh, eps=1e-5) > 0) != (h64[i] > 0)
        rows = flips.any(dim=2)                                                # [B, C]: rows containing a flipped element
        # a flip can only happen where the float64 value is within a few conv rounding errors (x rstd) of zero
        if flips.any():
            rstd = (y64[i].var(dim=2, unbiased=False, keepdim=True) + 1e-5).rsqrt()
            assert float((h64[i].abs() / (rstd * conv_err))[flips].max()) < 4.0
        total_flips += int(flips.sum())
        # rows fed (through later layers) only by un-flipped elements are exact; once a flip happened in block j, the blocks
        # before it (i < j) receive a perturbed upstream gradient everywhere, so only the LAST flipped block and the ones
        # after it can be checked row-wise
        err = (gh - g64[i]).abs().amax(dim=2) / float(g64[i].abs().max())
        if total_flips == int(flips.sum()):                                    # no flip in any later block
            assert float(err[~rows].max()) < 2e-5, (i, float(err[~rows].max()))
        print(f"down_{i + 1}: {int(flips.sum())} flipped of {flips.numel()} elements, rows affected {int(rows.sum())}, "
              f"max row err without flip {float(err[~rows].max()):.2e}, with {float(err[rows].max()) if rows.any() else 0.0:.2e}")
    assert total_flips <= 20, "sign flips must be isolated events (a systematic error would flip thousands)"


def test_inference_py_call_sequence_dataparallel_and_module_prefixed_ema(golden_dir):
    """inference.py:63-76: `model = nn.DataParallel(model, device_ids=cfg['devices'])`, then
    `model.load_state_dict(checkpoint['state_dict_ema'])` (keys carry the `module.` prefix), then `model(video_list)` with the
    loader's items.  The wrapped model must give the golden detections; train mode through the wrapper must return the loss dict."""
    g = np.load(os.path.join(golden_dir, "fwd_small.npz"), allow_pickle=True)
    model, sd, cfg = build(SMALL, int(g["wseed"]))
    fresh, _, _ = build(SMALL, 999)                                   # different weights: the checkpoint must replace them
    wrapped = torch.nn.DataParallel(fresh, device_ids=[0])
    wrapped.load_state_dict({"module." + k: v for k, v in sd.items()})
    wrapped.eval()
    lens = [int(v) for v in g["lens"]]
    x = synth_inputs(cfg, lens, int(g["xseed"]))
    for b, clip in enumerate(clips_from(x, lens)):
        res = wrapped([clip])[0]
        k = int(g["n_segments"][b])
        assert res["segments"].shape[0] == k
        assert sorted(res["kept_points"].tolist()) == sorted(g["kept_points"][b][:k].tolist())
        assert res["segments"].device.type == "cpu" and res["scores"].device.type == "cpu"
    wrapped.train()
    segs = [np.array([[10.0, 30.0]], np.float32), None]
    losses = wrapped(clips_from(x, lens, segs))
    assert set(losses) >= {"cls_loss", "reg_loss", "final_loss"} and torch.isfinite(losses["final_loss"]).item()
    losses["final_loss"].backward()


def test_eval_clip_longer_than_max_seq_len_matches_oracle():
    """av_fd_no_recon.py:458-463 + backbones.py:456-461: an eval clip longer than max_seq_len is padded to the next multiple of
    max_div_factor and the position table is linearly re-interpolated to that length (max_buffer_len_factor > 1 sizes the point
    generator for it, as in the reference).  Checked against the oracle evaluated the same way."""
    import torch.nn.functional as F
    from audio_visual_deepfake_detection_amd import build_model
    from oracle import ummaformer_ref as R
    cfg = R.make_cfg(**SMALL)
    sd = R.synth_state_dict(R.param_shapes(cfg), 61, True)
    model, _ = build_model("deepfake_exp10.yaml", video_input_dim=8, audio_input_dim=88, max_seq_len=192, embd_dim=32, fpn_dim=32,
                           head_dim=32, n_head=4, max_buffer_len_factor=2.0)
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    assert model.max_div_factor == 96                                   # stride 16 x window 6 at the coarsest local level
    T = 250                                                             # -> padded to 288
    x = torch.from_numpy(np.random.RandomState(62).standard_normal((96, T)).astype(np.float32))
    clip = dict(video_id="long.mp4", feats=x, fps=25.0, duration=T / 25.0, feat_stride=1.0, feat_num_frames=1.0)
    xt, mask = model.preprocessing([clip])
    assert xt.shape == (1, 288, 96) and float(mask.sum()) == T
    with torch.no_grad():
        logits, offsets, pmask, cls_scores, lens = model.forward_features(xt, mask)
    # oracle on the same padded length (it re-interpolates the 192-row position table like backbones.py:456-461)
    xo = F.pad(x, (0, 288 - T)).unsqueeze(0)
    mo = (torch.arange(288)[None, :] < T).float().unsqueeze(1)
    with torch.no_grad():
        out = R.model_forward(sd, cfg, xo, mo)
    ref_logits = torch.cat([z[0] for z in out["logits"]], 0)[:, 0]
    ref_offsets = torch.cat([z[0] for z in out["offsets"]], 0)
    assert maxerr(logits[0].cpu(), ref_logits) < TOL and maxerr(offsets[0].cpu(), ref_offsets) < TOL
    res = model([clip])[0]
    assert res["segments"].shape[1] == 2 and float(res["segments"].max()) <= T / 25.0 + 1e-6
