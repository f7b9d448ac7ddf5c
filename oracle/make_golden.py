"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference); never on the GPU box.
It imports the reference's libs.modeling / libs.utils (recipe: SURVEY.md appendix A),
loads the deterministic synthetic weights of oracle.ummaformer_ref.synth_state_dict into
the reference model, runs it, checks the CPU oracle against it, and stores only inputs
seeds + expected outputs (no reference source, no weights).

    python oracle/make_golden.py            # writes tests/golden/*.npz
"""
import copy
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ummaformer_ref as R  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.modules["torchsort"] = types.ModuleType("torchsort")   # losses.py:3 imports, never uses
    sys.modules["h5py"] = types.ModuleType("h5py")             # deepfake_video_audio.py:3 same
    from oracle.build_ref import build_ref_nms
    build_ref_nms()                                            # oracle/_ref/nms_1d_cpu*.so
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    from libs.core import load_config, config as C
    from libs.modeling import make_meta_arch
    return load_config, C, make_meta_arch


def ref_model_plain(load_config, C, make_meta_arch, cfg):
    """The reference's LocPointTransformer (meta_archs.py:162) with exp10's hyper-parameters; its constructor takes
    `input_dim`, which the reference's own config loader never fills, so the kwargs are assembled here."""
    import importlib
    importlib.import_module("libs.modeling.meta_archs")      # not imported by libs/modeling/__init__.py: registers it
    rc = load_config(f"{REF}/configs_train/deepfake_exp10.yaml", defaults=copy.deepcopy(C.DEFAULTS))
    m = dict(rc["model"])
    m.pop("video_input_dim")
    m.pop("mlp_ratio", None)
    m.update(backbone_type="conv" if cfg["variant"] == "plainconv" else "convTransformer", input_dim=cfg["input_dim"],
             audio_input_dim=0, max_seq_len=cfg["max_seq_len"],
             embd_dim=cfg["embd_dim"], fpn_dim=cfg["fpn_dim"], head_dim=cfg["head_dim"], n_head=cfg["n_head"])
    return make_meta_arch("LocPointTransformer", **m), rc


def ref_model(load_config, C, make_meta_arch, yaml_name, cfg):
    if cfg["variant"] in ("plain", "plainconv"):
        return ref_model_plain(load_config, C, make_meta_arch, cfg)
    rc = load_config(f"{REF}/configs_train/{yaml_name}", defaults=copy.deepcopy(C.DEFAULTS))
    m = rc["model"]
    vdim = cfg.get("video_dim", max(8, cfg["input_dim"] // 12))     # "video_dim": 0 = audio-only (BASELINE config 2)
    m.update(video_input_dim=vdim, audio_input_dim=cfg["input_dim"] - vdim,
             max_seq_len=cfg["max_seq_len"], embd_dim=cfg["embd_dim"], fpn_dim=cfg["fpn_dim"],
             head_dim=cfg["head_dim"], n_head=cfg["n_head"], use_rel_pe=bool(cfg.get("use_rel_pe", False)))
    model = make_meta_arch(rc["model_name"], **m)
    return model, rc


def synth_inputs(cfg, lens, seed):
    rs = np.random.RandomState(seed)
    B, C, T = len(lens), cfg["input_dim"], cfg["max_seq_len"]
    x = np.zeros((B, C, T), dtype=np.float32)
    for b, n in enumerate(lens):
        x[b, :, :n] = rs.standard_normal((C, n)).astype(np.float32)
    mask = (np.arange(T)[None, :] < np.asarray(lens)[:, None]).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(mask)[:, None, :]


def ref_forward_heads(model, x, mask_f):
    """Replays av_fd_no_recon.py:346-379 on pre-batched tensors (so B>1 works in eval)."""
    mb = mask_f > 0
    if type(model).__name__ == "PtTransformer" and type(model).__module__.endswith("meta_archs"):
        feats, masks = model.backbone(x, mb)                              # meta_archs.py:335-345
        fpn_feats, fpn_masks = model.neck(feats, masks)
        logits = [t.permute(0, 2, 1) for t in model.cls_head(fpn_feats, fpn_masks)]
        offsets = [t.permute(0, 2, 1) for t in model.reg_head(fpn_feats, fpn_masks)]
        return dict(logits=logits, offsets=offsets, cls_scores=torch.zeros(x.shape[0], 1), feats=feats, fpn=fpn_feats,
                    masks=[m.squeeze(1) for m in fpn_masks])
    front = model.interpolator if hasattr(model, "interpolator") else model.segmentandCls
    norm_inputs, reco, cls_scores = front(x, mb)
    recon = type(model).__module__.endswith("av_fd_meta_arch")        # exp5: Expansion output feeds the backbone
    feats, masks = model.backbone(x, norm_inputs, reco if recon else x, mb)
    fpn_feats, fpn_masks = model.neck(feats, masks)
    logits = [t.permute(0, 2, 1) for t in model.cls_head(fpn_feats, fpn_masks)]
    offsets = [t.permute(0, 2, 1) for t in model.reg_head(fpn_feats, fpn_masks)]
    return dict(logits=logits, offsets=offsets, cls_scores=cls_scores, feats=feats, fpn=fpn_feats,
                masks=[m.squeeze(1) for m in fpn_masks])


def maxdiff(a, b):
    return float((a - b).abs().max())


def match_segments(seg_a, sc_a, seg_b, sc_b):
    """Order-insensitive comparison of two detection lists: near-tied scores (|d| ~ 1e-8) may
    swap ranks between two fp32 implementations (the reference's own sort is unstable,
    av_fd_no_recon.py:789), so each row of A is matched to its nearest row of B."""
    if seg_a.shape[0] == 0:
        return 0.0, 0.0, torch.zeros(0, dtype=torch.int64)
    d = (seg_a[:, None, :] - seg_b[None, :, :]).abs().amax(-1) + (sc_a[:, None] - sc_b[None, :]).abs()
    perm = d.argmin(dim=1)
    assert perm.unique().numel() == perm.numel(), "segment lists do not match one-to-one"
    return maxdiff(seg_a, seg_b[perm]), maxdiff(sc_a, sc_b[perm]), perm


def yaml_for(cfg):
    return {"norecon": "deepfake_exp10.yaml", "the": "deepfake_exp13.yaml", "recon": "deepfake_exp5.yaml",
            "plain": "deepfake_exp10.yaml", "plainconv": "deepfake_exp10.yaml"}[cfg["variant"]]


def eval_clips(name, model, sd, cfg, clips):
    """`model.eval()([clip])` of the reference per clip (forward -> decode -> its C++ NMS -> seconds), the oracle's
    `infer_clip` asserted against it, and the padded arrays a fixture stores (segments, scores, kept point index of each
    REFERENCE segment -- taken from the oracle after the one-to-one match)."""
    segs_all, scores_all, n_all, vc_all, kept_all = [], [], [], [], []
    for b, clip in enumerate(clips):
        with torch.no_grad():
            res = model([clip])[0]
            o = R.infer_clip(sd, cfg, clip["feats"], dict(fps=clip["fps"], duration=clip["duration"],
                                                          feat_stride=clip["feat_stride"],
                                                          feat_num_frames=clip["feat_num_frames"]))
        k = res["segments"].shape[0]
        assert o["segments"].shape[0] == k, (o["segments"].shape, k)
        ds, dsc, perm = match_segments(res["segments"], res["scores"], o["segments"], o["scores"])
        print(f"[{name}] clip {b}: {k} segments, oracle-vs-ref seg diff {ds:.3e} score diff {dsc:.3e} "
              f"(rank moves among near-ties: {int((perm != torch.arange(k)).sum())})")
        assert ds < 2e-3 and dsc < 1e-4
        kp = np.full((cfg["max_seg_num"],), -1, np.int64)
        kp[:k] = o["kept_points"][perm].numpy()       # point index of each REFERENCE segment
        kept_all.append(kp)
        pad = np.zeros((cfg["max_seg_num"], 2), np.float32)
        pad[:k] = res["segments"].numpy()
        sp = np.zeros((cfg["max_seg_num"],), np.float32)
        sp[:k] = res["scores"].numpy()
        segs_all.append(pad)
        scores_all.append(sp)
        n_all.append(k)
        vc_all.append(res["video_cls"].numpy() if "video_cls" in res else np.zeros(1, np.float32))
    return dict(segments=np.stack(segs_all), scores=np.stack(scores_all), n_segments=np.asarray(n_all),
                video_cls=np.stack(vc_all), kept_points=np.stack(kept_all))


def case_forward(name, cfg, lens, wseed, xseed, ctx, trained_like=True, store_levels=True):
    load_config, C, make_meta_arch = ctx
    model, _ = ref_model(load_config, C, make_meta_arch, yaml_for(cfg), cfg)
    shapes = R.param_shapes(cfg)
    ref_sd = model.state_dict()
    assert set(shapes) == set(ref_sd), (set(shapes) ^ set(ref_sd))
    for k in shapes:
        assert tuple(ref_sd[k].shape) == tuple(shapes[k]), (k, ref_sd[k].shape, shapes[k])
    sd = R.synth_state_dict(shapes, wseed, trained_like)
    model.load_state_dict(sd, strict=True)
    model.eval()
    x, mask = synth_inputs(cfg, lens, xseed)
    with torch.no_grad():
        ref = ref_forward_heads(model, x, mask)
        mine = R.model_forward(sd, cfg, x, mask)
    dl = max(maxdiff(a, b) for a, b in zip(ref["logits"], mine["logits"]))
    do = max(maxdiff(a, b) for a, b in zip(ref["offsets"], mine["offsets"]))
    dc = 0.0 if mine["cls_scores"] is None else maxdiff(ref["cls_scores"], mine["cls_scores"])
    df = max(maxdiff(a, b) for a, b in zip(ref["fpn"], mine["fpn"]))
    print(f"[{name}] oracle vs reference: logits {dl:.3e} offsets {do:.3e} cls {dc:.3e} fpn {df:.3e}")
    assert dl < 2e-4 and do < 2e-4 and dc < 2e-4, "oracle restatement disagrees with the reference"
    out = dict(
        cfg_keys=np.array(sorted(cfg.keys())), lens=np.asarray(lens), wseed=wseed, xseed=xseed,
        trained_like=int(trained_like),
        logits=torch.cat(ref["logits"], 1).numpy(), offsets=torch.cat(ref["offsets"], 1).numpy(),
        cls_scores=ref["cls_scores"].numpy(),
        fpn_sum=np.array([float(f.double().sum()) for f in ref["fpn"]]),
        fpn_abs=np.array([float(f.double().abs().sum()) for f in ref["fpn"]]),
    )
    # full eval path of the reference (B=1 per clip): forward -> decode -> C++ NMS -> seconds
    clips = [dict(video_id=f"c{b}.mp4", feats=x[b, :, :n].clone(), fps=25.0, duration=n / 25.0, feat_stride=1.0,
                  feat_num_frames=1.0) for b, n in enumerate(lens)]
    out.update(eval_clips(name, model, sd, cfg, clips))
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **out)
    return model, sd, x, mask


def case_train(name, cfg, lens, wseed, xseed, ctx):
    """Training-mode losses + a few gradients with drop-path / dropout disabled
    (drop_prob=0 -> drop_path is the identity, blocks.py:1401-1402)."""
    load_config, C, make_meta_arch = ctx
    model, _ = ref_model(load_config, C, make_meta_arch, yaml_for(cfg), cfg)
    sd = R.synth_state_dict(R.param_shapes(cfg), wseed, True)
    model.load_state_dict(sd, strict=True)
    model.train()
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    if hasattr(model, "interpolator"):
        model.interpolator.dp1.p = 0.0
    the = cfg["variant"] == "the"
    plain = cfg["variant"] in ("plain", "plainconv")
    x, mask = synth_inputs(cfg, lens, xseed)
    rs = np.random.RandomState(xseed + 7)
    vids, gts, gls, gfl = [], [], [], []
    for b, n in enumerate(lens):
        if b % 2 == 0 or plain:               # LocPointTransformer trains on clips with segments only
            k = 1 + b % 3
            st = np.sort(rs.uniform(0, n - 20, size=k)).astype(np.float32)
            ln = rs.uniform(2.0, 18.0, size=k).astype(np.float32)
            seg = torch.from_numpy(np.stack([st, st + ln], 1))
            lab = torch.zeros(k, dtype=torch.int64)
        else:
            seg, lab = None, None
        gts.append(seg)
        gls.append(lab)
        fl = None
        if seg is not None:
            fl = torch.zeros(cfg["max_seq_len"])
            for a_, b_ in seg.tolist():
                fl[int(a_):int(b_) + 1] = 1.0
        gfl.append(fl)
        vids.append(dict(video_id=f"c{b}", feats=x[b, :, :n].clone(), segments=seg, labels=lab, fps=25.0,
                         duration=n / 25.0, feat_stride=1.0, feat_num_frames=1.0, gt_frame_labels=fl))
    losses = model(vids)
    losses["final_loss"].backward()
    # oracle
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = R.model_forward(sdg, cfg, x, mask, training=True)
    ol, norm, npos = R.losses(cfg, out, gts, gls, cfg["init_loss_norm"], x_in=x,
                              gt_frame_labels=gfl if the else None)
    ol["final_loss"].backward()
    recon = cfg["variant"] == "recon"
    lkeys = (("cls_loss", "reg_loss", "final_loss") if plain else ("cls_loss", "reg_loss", "reco_cls_loss", "final_loss")) + \
        (("reco_seg_loss",) if the else ()) + (("reco_loss",) if recon else ())
    for k in lkeys:
        print(f"[{name}] {k}: ref {float(losses[k]):.6f} oracle {float(ol[k]):.6f}")
        assert abs(float(losses[k]) - float(ol[k])) < 1e-4 * max(1.0, abs(float(losses[k])))
    assert abs(norm - model.loss_normalizer) < 1e-6
    gkeys = ["backbone.embd.0.conv.weight", "backbone.stem.0.attn.query.weight",
             "backbone.branch.4.attn.value_conv.conv.weight", "backbone.lh_branch.2.mlp.3.weight",
             "backbone.hh_branch.0.attn.key_norm.weight", "backbone.branch.1.drop_path_attn.scale",
             "neck.fpn_convs.3.conv.weight", "cls_head.cls_head.conv.bias", "reg_head.scale.2.scale",
             "reg_head.offset_head.conv.weight"]
    if cfg["variant"] == "plainconv":
        gkeys = ["backbone.embd.0.conv.weight", "backbone.stem.0.conv1.conv.weight", "backbone.stem.1.conv2.conv.bias",
                 "backbone.branch.2.downsample.conv.weight", "backbone.branch.4.conv1.conv.weight", "neck.fpn_convs.3.conv.weight",
                 "cls_head.cls_head.conv.bias", "reg_head.scale.2.scale", "reg_head.offset_head.conv.weight"]
    elif plain:
        gkeys = [k for k in gkeys if "lh_branch" not in k and "hh_branch" not in k] + ["backbone.stem.1.mlp.0.weight"]
    elif the:
        gkeys += ["segmentandCls.contraction.down_3.conv_block.conv.weight", "segmentandCls.conv0.0.weight",
                  "segmentandCls.seg_linear.weight", "segmentandCls.cls_linear1.weight",
                  "segmentandCls.cls_linear1.bias", "segmentandCls.contraction.down_1.conv_block.conv.bias"]
    else:
        gkeys += ["interpolator.contraction.down_3.conv_block.conv.weight", "interpolator.conv1.weight",
                  "interpolator.bn1.bias"]
    if cfg.get("use_rel_pe"):
        gkeys += ["backbone.stem.1.attn.rel_pe", "backbone.branch.2.attn.rel_pe", "backbone.lh_branch.0.attn.rel_pe"]
    if recon:
        gkeys += ["interpolator.expansion.up_1.conv_transpose.conv.weight",
                  "interpolator.expansion.up_3.conv_transpose.conv.bias",
                  "interpolator.expansion.up_5.conv_transpose.conv.weight",
                  "backbone.resselfattention.attn.key_conv.conv.weight", "backbone.resselfattention.lnk.weight"]
    named = dict(model.named_parameters())
    store = {}
    worst = 0.0
    for k in gkeys:
        g = named[k].grad
        go = sdg[k].grad
        rel = float((g - go).abs().max() / (g.abs().max() + 1e-12))
        worst = max(worst, rel)
        flat = g.flatten()
        stride = max(1, flat.numel() // 4096)
        store["grad::" + k] = flat[::stride][:4096].numpy().copy()   # strided sample (fixtures stay small)
        store["gsum::" + k] = np.array([float(flat.double().sum()), float(flat.double().abs().sum())])
    print(f"[{name}] worst relative grad diff oracle vs ref: {worst:.3e}")
    assert worst < 2e-3
    dead = sorted(k for k, p in named.items() if p.grad is None)
    np.savez_compressed(
        os.path.join(GOLD, f"{name}.npz"), lens=np.asarray(lens), wseed=wseed, xseed=xseed,
        gt_segments=np.array([np.zeros((0, 2), np.float32) if g is None else g.numpy() for g in gts],
                             dtype=object),
        is_fake=np.array([g is not None for g in gts]),
        losses=np.array([float(losses[k]) if k in losses else 0.0 for k in ("cls_loss", "reg_loss", "reco_cls_loss", "final_loss")]),
        reco_seg_loss=float(losses["reco_seg_loss"]) if the else 0.0,
        reco_loss=float(losses["reco_loss"]) if recon else 0.0,
        gt_frame_labels=np.array([np.zeros(0, np.float32) if f is None else f.numpy() for f in gfl], dtype=object),
        loss_normalizer=float(model.loss_normalizer), num_dead=len(dead),
        dead_prefixes=np.array(sorted({".".join(d.split(".")[:3]) for d in dead})), **store)


def case_nms():
    import nms_1d_cpu
    from libs.utils import batched_nms
    rs = np.random.RandomState(5)
    cases = {}
    for i, n in enumerate([0, 1, 2, 7, 64, 300, 1512]):
        st = rs.uniform(0, 700, n).astype(np.float32)
        ln = rs.uniform(0.01, 40, n).astype(np.float32)
        segs = np.stack([st, st + ln], 1).astype(np.float32).reshape(n, 2)
        sc = rs.uniform(0.0, 1.0, n).astype(np.float32)
        if n >= 64:                       # exact score ties and duplicate segments
            sc[10:20] = sc[10]
            segs[30:34] = segs[30]
        keep = nms_1d_cpu.nms(torch.from_numpy(segs), torch.from_numpy(sc), 0.1).numpy() if n else np.zeros(0, np.int64)
        cases[f"segs{i}"], cases[f"scores{i}"], cases[f"keep{i}"] = segs, sc, keep
        for tag, ms in (("a", 0.001), ("b", 0.2)):
            s, c, l = batched_nms(torch.from_numpy(segs), torch.from_numpy(sc), torch.zeros(n, dtype=torch.int64),
                                  0.1, ms, 100, use_soft_nms=False, multiclass=False, sigma=0.75, voting_thresh=0.9)
            cases[f"bn_segs{i}{tag}"], cases[f"bn_scores{i}{tag}"] = s.numpy(), c.numpy()
    cases["n_cases"] = 7
    np.savez_compressed(os.path.join(GOLD, "nms.npz"), **cases)
    print("[nms] stored", cases["n_cases"], "cases")


def case_nms_modes():
    """The reference's batched_nms (libs/utils/nms.py:103-190, native nms_1d_cpu incl. softnms) in its soft /
    multiclass branches, on several classes with duplicated scores."""
    from libs.utils.nms import batched_nms
    out = {}
    rs = np.random.RandomState(77)
    ci = 0
    for n in (0, 1, 7, 60, 300):
        st = rs.uniform(0, 100, n).astype(np.float32)
        segs = np.stack([st, st + rs.uniform(0.5, 20, n).astype(np.float32)], 1).reshape(-1, 2)
        sc = np.round(rs.uniform(0, 1, n), 2).astype(np.float32)
        cls = rs.randint(0, 3, n).astype(np.int64)
        for soft, multi in ((True, True), (True, False), (False, True)):
            s, c, l = batched_nms(torch.from_numpy(segs), torch.from_numpy(sc), torch.from_numpy(cls), 0.1, 0.001, 100,
                                  use_soft_nms=soft, multiclass=multi, sigma=0.5, voting_thresh=0.75)
            out[f"c{ci}_in"] = np.concatenate([segs, sc[:, None], cls[:, None].astype(np.float32)], 1)
            out[f"c{ci}_mode"] = np.array([int(soft), int(multi)])
            out[f"c{ci}_segs"], out[f"c{ci}_scores"], out[f"c{ci}_cls"] = s.numpy(), c.numpy(), l.numpy()
            ci += 1
    out["n_cases"] = np.array(ci)
    np.savez_compressed(os.path.join(GOLD, "nms_modes.npz"), **out)
    print("[nms_modes] stored", ci, "cases")


def case_interp():
    import torch.nn.functional as F
    out = {}
    # known answer quoted in the reference source (deepfake_video_audio.py:1726)
    ka = F.interpolate(torch.tensor([[[1.0, 2.0, 3.0, 4.0]]]), size=8, mode="linear", align_corners=False)
    assert torch.allclose(ka[0, 0], torch.tensor([1.0, 1.25, 1.75, 2.25, 2.75, 3.25, 3.75, 4.0]))
    out["known_in"] = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    out["known_out"] = ka[0, 0].numpy()
    rs = np.random.RandomState(11)
    for i, (tv, ta, te) in enumerate([(216, 107, 431), (768, 383, 1535), (5, 1, 3), (900, 12, 2000)]):
        v = rs.standard_normal((tv, 6)).astype(np.float32)
        a = rs.standard_normal((ta, 10)).astype(np.float32)
        e = rs.standard_normal((te, 4)).astype(np.float32)
        feats = R.interp_concat([v, a, e], 768)
        # the same with the reference's literal op sequence (deepfake_video_audio.py:1722-1756)
        vv = torch.from_numpy(np.ascontiguousarray(v.transpose()))
        if vv.shape[-1] != 768:
            vv = F.interpolate(vv.unsqueeze(0), size=768, mode="linear", align_corners=False).squeeze(0)
        aa = F.interpolate(torch.from_numpy(np.ascontiguousarray(a.transpose())).unsqueeze(0), size=768,
                           mode="linear", align_corners=False).squeeze(0)
        ee = F.interpolate(torch.from_numpy(np.ascontiguousarray(e.transpose())).unsqueeze(0), size=768,
                           mode="linear", align_corners=False).squeeze(0)
        ref = torch.cat([vv, aa, ee], 0)
        assert torch.equal(ref, feats)
        out[f"v{i}"], out[f"a{i}"], out[f"e{i}"], out[f"out{i}"] = v, a, e, ref.numpy()
    out["n_cases"] = 4
    np.savez_compressed(os.path.join(GOLD, "interp.npz"), **out)
    print("[interp] stored 4 cases + known answer")


def case_datasets():
    """Reference dataset classes on a synthetic tree (oracle/synth_data.py): per-item metadata, segments after
    the seeded random crop, and checksums of the interpolated+concatenated+cropped feature matrix."""
    import random
    import tempfile
    from oracle.synth_data import make_synth_tree, dataset_kwargs
    from libs.datasets import make_dataset, make_inference_dataset
    root = tempfile.mkdtemp(prefix="avdf_synth_")
    rels = make_synth_tree(root, seed=0)
    kw = dataset_kwargs(root)
    out = {"n": len(rels)}
    for tag, name in (("a", "deepfake_video_audioEmoBYOLA"), ("t", "deepfake_video_audioEmoBYOLA_THE")):
        ds = make_dataset(name, True, ["train"], **kw)
        for i in range(len(ds)):
            random.seed(100 + i)
            d = ds[i]
            f = d["feats"]
            out[f"{tag}{i}_shape"] = np.array(f.shape)
            out[f"{tag}{i}_sum"] = np.array([float(f.double().sum()), float(f.double().abs().sum())])
            out[f"{tag}{i}_col0"] = f[:, 0].numpy()
            out[f"{tag}{i}_colN"] = f[:, -1].numpy()
            out[f"{tag}{i}_segments"] = np.zeros((0, 2), np.float32) if d["segments"] is None else d["segments"].numpy()
            out[f"{tag}{i}_meta"] = np.array([d["fps"], d["duration"], d["feat_stride"], d["feat_num_frames"]], np.float64)
            out[f"{tag}{i}_av"] = d["av_labels"].numpy()
            if tag == "t":
                gl = d["gt_frame_labels"]
                out[f"{tag}{i}_frames"] = np.zeros(0, np.float32) if gl is None else gl.numpy()
    kwi = dict(crop_ratio=kw["crop_ratio"], default_fps=None, downsample_rate=0, video_feat_folder=kw["video_feat_folder"],
               audio_feat_folder=None, audio_byola_feat_folder=kw["audio_byola_feat_folder"],
               audio_emo_feat_folder=kw["audio_emo_feat_folder"], audio_file_ext=".npy", num_classes=1, input_dim=0,
               video_input_dim=8, audio_input_dim=24, feat_stride=1, num_frames=1,
               test_folder=os.path.join(root, "test_folder"), trunc_thresh=0.5, max_seq_len=768, force_upsampling=True)
    dsi = make_inference_dataset("deepfake_video_audioEmoBYOLA_inference", False, ["test"], 1, **kwi)
    for i in range(len(dsi)):
        d = dsi[i]
        f = d["feats"]
        out[f"i{i}_sum"] = np.array([float(f.double().sum()), float(f.double().abs().sum())])
        out[f"i{i}_meta"] = np.array([d["fps"], d["duration"], d["feat_stride"], d["feat_num_frames"]], np.float64)
    np.savez_compressed(os.path.join(GOLD, "datasets.npz"), **out)
    print("[datasets] stored", len(rels), "items x 3 dataset classes")


def synth_detections(seed, nvid=12, nlab=2):
    """Ground truth + predictions tables for the evaluator fixtures: a few videos without predictions / without GT,
    duplicated scores (tie order) and near-threshold overlaps."""
    rs = np.random.RandomState(seed)
    gt = {"video-id": [], "t-start": [], "t-end": [], "label": []}
    pr = {"video-id": [], "t-start": [], "t-end": [], "label": [], "score": []}
    for v in range(nvid):
        vid = f"v{v:03d}.mp4"
        ng = int(rs.randint(0, 4)) if v else 2
        segs = []
        for _ in range(ng):
            s = float(rs.uniform(0, 25)); e = s + float(rs.uniform(0.2, 3.0))
            segs.append((s, e))
            gt["video-id"].append(vid); gt["t-start"].append(s); gt["t-end"].append(e); gt["label"].append(int(rs.randint(0, nlab)) * 3 + 2)
        if v % 5 == 4:
            continue                                   # video without predictions
        for (s, e) in segs:
            for _ in range(int(rs.randint(1, 4))):
                j = rs.uniform(-0.6, 0.6, 2)
                pr["video-id"].append(vid); pr["t-start"].append(s + j[0]); pr["t-end"].append(max(s + j[0] + 0.05, e + j[1]))
                pr["label"].append(int(rs.randint(0, nlab)) * 3 + 2); pr["score"].append(float(np.round(rs.uniform(0, 1), 2)))
        for _ in range(int(rs.randint(0, 3))):
            s = float(rs.uniform(0, 28))
            pr["video-id"].append(vid); pr["t-start"].append(s); pr["t-end"].append(s + float(rs.uniform(0.1, 2)))
            pr["label"].append(int(rs.randint(0, nlab)) * 3 + 2); pr["score"].append(float(np.round(rs.uniform(0, 1), 2)))
    pr["video-id"].append("ghost.mp4"); pr["t-start"].append(1.0); pr["t-end"].append(2.0); pr["label"].append(2); pr["score"].append(0.5)
    return gt, pr


def case_metrics():
    """Reference evaluator (libs/utils/metrics.py:112-447) on synthetic tables.  The reference still calls `np.float`
    (metrics.py:326-327), gone since numpy 1.24: the alias is restored HERE (generator process only)."""
    import json
    import tempfile
    import pandas as pd
    if not hasattr(np, "float"):
        np.float = float
    from libs.utils.metrics import ANETdetection
    out = {}
    for ci, seed in enumerate((11, 12, 13)):
        gt, pr = synth_detections(seed)
        db = {}
        for v, s, e, l in zip(gt["video-id"], gt["t-start"], gt["t-end"], gt["label"]):
            db.setdefault(v, {"subset": "test", "annotations": []})["annotations"].append({"segment": [s, e], "label_id": l})
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump({"database": db}, f)
        ev = ANETdetection(f.name, "test", tiou_thresholds=np.array([0.5, 0.75, 0.95]), top_k=(1, 5), num_workers=1)
        mAP, avg, mRecall = ev.evaluate(pd.DataFrame(pr), verbose=False)
        os.unlink(f.name)
        out[f"c{ci}_seed"] = np.array(seed)
        out[f"c{ci}_mAP"], out[f"c{ci}_avg"], out[f"c{ci}_mRecall"], out[f"c{ci}_ap"] = mAP, np.array(avg), mRecall, ev.ap
    np.savez_compressed(os.path.join(GOLD, "metrics.npz"), **out)
    print("[metrics] stored 3 evaluator cases; mAP of the last:", mAP)


def case_lr_schedules():
    """lr held by the reference's optimiser at every iteration: make_scheduler (train_utils.py:147-215) stepped per
    iteration like train_one_epoch does (train_utils.py:316)."""
    from libs.utils.train_utils import make_scheduler
    out = {}
    cases = [dict(learning_rate=1e-3, epochs=15, warmup=True, warmup_epochs=5, schedule_type="cosine", schedule_steps=[], schedule_gamma=0.1),
             dict(learning_rate=2e-4, epochs=6, warmup=True, warmup_epochs=2, schedule_type="multistep", schedule_steps=[2, 4], schedule_gamma=0.1),
             dict(learning_rate=5e-4, epochs=4, warmup=False, warmup_epochs=0, schedule_type="cosine", schedule_steps=[], schedule_gamma=0.1)]
    for ci, oc in enumerate(cases):
        nit = 7
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=oc["learning_rate"])
        sch = make_scheduler(opt, oc, nit)
        total = (oc["epochs"] + (oc["warmup_epochs"] if oc["warmup"] else 0)) * nit
        lrs = []
        for _ in range(total):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"c{ci}_lr"] = np.asarray(lrs, dtype=np.float64)
        out[f"c{ci}_cfg"] = np.array([oc["learning_rate"], oc["epochs"], float(oc["warmup"]), oc["warmup_epochs"],
                                      0.0 if oc["schedule_type"] == "cosine" else 1.0, nit] + list(oc["schedule_steps"]), dtype=np.float64)
    np.savez_compressed(os.path.join(GOLD, "lr_schedules.npz"), **out)
    print("[lr] stored", len(cases), "schedules")


SMALL = dict(input_dim=96, max_seq_len=192, embd_dim=32, n_head=4, fpn_dim=32, head_dim=32)
MID = dict(input_dim=192, max_seq_len=384, embd_dim=128, n_head=4, fpn_dim=128, head_dim=128)


def case_byola_net():
    """BYOL-A network: the reference's own AudioNTT2020Task6 (audio_feature/content_audio/byol_a/models.py:48-85, a file
    that imports only torch -- loaded by path so that byol_a/common.py's torchaudio import is never reached) on seeded
    weights and a seeded log-mel batch; asserts the oracle restatement against it and stores the reference output."""
    import importlib.util
    from oracle import byola_ref as BR
    spec = importlib.util.spec_from_file_location("byol_a_models_ref", os.path.join(REF, "audio_feature", "content_audio", "byol_a", "models.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    wseed, xseed = 4242, 4243
    net = mod.AudioNTT2020Task6(n_mels=64, d=2048).eval()
    sd = BR.seeded_state_dict(wseed)
    missing = net.load_state_dict(sd, strict=True)
    lms = torch.from_numpy(np.random.RandomState(xseed).randn(2, 1, 64, 203).astype(np.float32))
    with torch.no_grad():
        ref = net(lms)
        mine = BR.byola_forward(sd, lms)
    err = float((ref - mine).abs().max() / ref.abs().max())
    print(f"byola_net: reference out {tuple(ref.shape)}, oracle vs reference rel err {err:.2e}")
    assert err < 1e-6 and ref.shape == (2, 25, 2048)
    np.savez_compressed(os.path.join(GOLD, "byola_net.npz"), wseed=wseed, xseed=xseed, lms_shape=np.array(lms.shape),
                        out=ref.numpy())


AUDIO_EXP10 = dict(input_dim=2048, video_dim=0)                       # BASELINE config 2: BYOL-A features only, exp10 architecture
EXP5 = dict(variant="recon", input_dim=1024, video_dim=256)           # BASELINE config 1: deepfake_exp5.yaml at full size


def case_tinydataset(ctx):
    """BASELINE config 1: the reference's `tinydataset/metadata/**.json` (12 files) -> its `deepfake_video_audio` dataset
    class (libs/datasets/deepfake_video_audio.py:739-1102) on seeded synthetic features of the extractors' shapes -> its
    exp5 model (`AVLocPointTransformerRecoveryNoNorm`, full size, seeded weights) in eval mode, one clip per call like
    inference.py.  Stores the metadata fields the datasets read (data), per-item checksums / segments and the detections."""
    import glob
    import json
    import random
    import tempfile
    from oracle.synth_data import TINY_FIELDS, tiny_tree_from_fixture, tiny_dataset_kwargs
    from libs.datasets import make_dataset
    load_config, C, make_meta_arch = ctx
    metas = []
    for f in sorted(glob.glob(f"{REF}/tinydataset/metadata/**/*.json", recursive=True)):
        v = json.load(open(f))
        assert v["file"].replace(".mp4", ".json") == os.path.relpath(f, f"{REF}/tinydataset/metadata")
        metas.append({k: v[k] for k in TINY_FIELDS})
    assert len(metas) == 12
    fixture = os.path.join(GOLD, "tinydataset_meta.json")
    with open(fixture, "w") as f:
        json.dump(metas, f, indent=0)
    root = tempfile.mkdtemp(prefix="avdf_tiny_")
    rels = tiny_tree_from_fixture(root, fixture, seed=500)
    kw = tiny_dataset_kwargs(root)
    kw_ref = dict(kw, json_folder=f"{REF}/tinydataset/metadata")           # the reference reads its OWN files
    out = {"n": len(rels), "wseed": 71}
    cfg = R.make_cfg(**EXP5)
    model, _ = ref_model(load_config, C, make_meta_arch, "deepfake_exp5.yaml", cfg)
    sd = R.synth_state_dict(R.param_shapes(cfg), 71, True)
    model.load_state_dict(sd, strict=True)
    model.eval()
    ds_eval = make_dataset("deepfake_video_audio", False, ["train"], **kw_ref)
    ds_train = make_dataset("deepfake_video_audio", True, ["train"], **kw_ref)
    clips = []
    for i in range(len(ds_eval)):
        d = ds_eval[i]
        f = d["feats"]
        assert tuple(f.shape) == (1024, 768)
        out[f"e{i}_sum"] = np.array([float(f.double().sum()), float(f.double().abs().sum())])
        out[f"e{i}_row0"], out[f"e{i}_rowN"] = f[0].numpy(), f[-1].numpy()
        out[f"e{i}_segments"] = np.zeros((0, 2), np.float32) if d["segments"] is None else d["segments"].numpy()
        out[f"e{i}_meta"] = np.array([d["fps"], d["duration"], d["feat_stride"], d["feat_num_frames"]], np.float64)
        out[f"e{i}_av"] = d["av_labels"].numpy()
        clips.append(d)
        random.seed(900 + i)
        t = ds_train[i]
        ft = t["feats"]
        out[f"t{i}_shape"] = np.array(ft.shape)
        out[f"t{i}_sum"] = np.array([float(ft.double().sum()), float(ft.double().abs().sum())])
        out[f"t{i}_segments"] = np.zeros((0, 2), np.float32) if t["segments"] is None else t["segments"].numpy()
    out.update(eval_clips("tinydataset", model, sd, cfg, clips))
    np.savez_compressed(os.path.join(GOLD, "tinydataset.npz"), **out)
    print("[tinydataset] stored", len(rels), "items; segments per clip:", out["n_segments"].tolist())


def case_optimizer_split(ctx):
    """make_optimizer's decay / no-decay partition (libs/utils/train_utils.py:65-122) of the three shipped models: counts
    and a digest of the sorted decayed names (exp10 = 195 decayed / 413 not)."""
    import hashlib
    from libs.utils.train_utils import make_optimizer
    load_config, C, make_meta_arch = ctx
    out = {}
    for tag, yaml_name in (("exp10", "deepfake_exp10.yaml"), ("exp5", "deepfake_exp5.yaml"), ("exp13", "deepfake_exp13.yaml")):
        rc = load_config(f"{REF}/configs_train/{yaml_name}", defaults=copy.deepcopy(C.DEFAULTS))
        model = make_meta_arch(rc["model_name"], **rc["model"])
        opt = make_optimizer(model, rc["opt"])
        names = {id(p): n for n, p in model.named_parameters()}
        decay = sorted(names[id(p)] for p in opt.param_groups[0]["params"])
        no_decay = sorted(names[id(p)] for p in opt.param_groups[1]["params"])
        assert opt.param_groups[1]["weight_decay"] == 0.0
        out[f"{tag}_counts"] = np.array([len(decay), len(no_decay)])
        out[f"{tag}_decay_sha"] = np.array(hashlib.sha256("\n".join(decay).encode()).hexdigest())
        out[f"{tag}_nodecay_sha"] = np.array(hashlib.sha256("\n".join(no_decay).encode()).hexdigest())
        out[f"{tag}_hparams"] = np.array([opt.param_groups[0]["weight_decay"], opt.param_groups[0]["lr"]])
        print(f"[optimizer_split] {tag}: {len(decay)} decayed / {len(no_decay)} not")
    assert tuple(out["exp10_counts"]) == (195, 413)
    np.savez_compressed(os.path.join(GOLD, "optimizer_split.npz"), **out)


def import_emotion2vec_torch_only():
    """upstream/models/{modules,base}.py of audio_feature/emotion_audio imported by path into a synthetic package.
    Names they import at module level from packages that are absent here -- and that the functions pinned below NEVER
    execute -- are stubbed: fairseq.modules.{LayerNorm, SamePad, TransposeLast, GradMultiply},
    fairseq.data.data_utils.compute_mask_indices, fairseq.utils.index_put, omegaconf.{MISSING, II} (dataclass field
    defaults).  Executed for the fixtures: get_alibi, get_alibi_bias (base.py:538-642) and AltAttention (modules.py:252-322),
    all torch-only.  AltBlock stays unpinned: its constructor takes Mlp / DropPath from timm, which would be executed."""
    import importlib.util
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    unused = type("NeverExecuted", (), {})
    stub("fairseq")
    stub("fairseq.modules", LayerNorm=unused, SamePad=unused, TransposeLast=unused, GradMultiply=unused)
    stub("fairseq.data")
    stub("fairseq.data.data_utils", compute_mask_indices=unused)
    stub("fairseq.utils", index_put=unused)
    stub("omegaconf", MISSING="???", II=lambda s: s)
    root = os.path.join(REF, "audio_feature", "emotion_audio", "upstream", "models")
    pkg = types.ModuleType("e2v_upstream_models")
    pkg.__path__ = [root]
    sys.modules["e2v_upstream_models"] = pkg
    mods = {}
    for name in ("modules", "base"):
        spec = importlib.util.spec_from_file_location(f"e2v_upstream_models.{name}", os.path.join(root, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"e2v_upstream_models.{name}"] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["modules"], mods["base"]


def case_emotion2vec_parts():
    """ALiBi bias + attention of Emotion2Vec pinned to the reference's own torch-only code (see the import helper)."""
    from oracle import emotion2vec_ref as ER
    sys.dont_write_bytecode = True
    M, Bm = import_emotion2vec_torch_only()
    out = {}
    for heads, T in ((12, 33), (8, 17), (6, 9)):
        ref = Bm.get_alibi(T, heads)
        mine = ER.alibi_bias(T, heads)
        assert torch.equal(ref, mine), (heads, T)
        out[f"alibi_h{heads}_t{T}"] = ref.numpy()
    cache = {}
    b = Bm.get_alibi_bias(cache, batch_size=2, time_steps=21, heads=12, dtype=torch.float32, device=torch.device("cpu"))
    assert b.shape == (2, 12, 21, 21) and torch.equal(b[0], ER.alibi_bias(21, 12)) and torch.equal(b[1], b[0])
    b2 = Bm.get_alibi_bias(cache, batch_size=1, time_steps=13, heads=12, dtype=torch.float32, device=torch.device("cpu"))
    assert torch.equal(b2[0], ER.alibi_bias(13, 12))             # served from the cached, larger table
    out["alibi_bias_b2_t21"] = b.numpy()
    # AltAttention at the model's real width (768, 12 heads), seeded weights and input, with a per-head scaled ALiBi bias
    dim, heads, N, wseed, xseed = 768, 12, 57, 8101, 8102
    rs = np.random.RandomState(wseed)
    attn = M.AltAttention(dim, num_heads=heads, qkv_bias=True).eval()
    sd = {"qkv.weight": rs.uniform(-1, 1, (3 * dim, dim)) / np.sqrt(dim) * 2.0, "qkv.bias": 0.1 * rs.standard_normal(3 * dim),
          "proj.weight": rs.uniform(-1, 1, (dim, dim)) / np.sqrt(dim), "proj.bias": 0.1 * rs.standard_normal(dim)}
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, v in sd.items()}
    attn.load_state_dict(sd, strict=True)
    x = torch.from_numpy(np.random.RandomState(xseed).standard_normal((2, N, dim)).astype(np.float32))
    scale = torch.from_numpy((0.5 + rs.uniform(0, 1, heads)).astype(np.float32)).view(1, heads, 1, 1)
    bias = Bm.get_alibi_bias({}, 2, N, heads, torch.float32, torch.device("cpu")) * scale
    with torch.no_grad():
        ref_b = attn(x, None, bias.clone())
        ref_n = attn(x, None, None)
        mine_b = ER.alt_attention({"attn." + k: v for k, v in sd.items()}, "", x, heads, bias[0])
        mine_n = ER.alt_attention({"attn." + k: v for k, v in sd.items()}, "", x, heads, None)
    e1, e2 = maxdiff(ref_b, mine_b), maxdiff(ref_n, mine_n)
    print(f"[emotion2vec_parts] AltAttention oracle vs reference: with alibi {e1:.2e}, without {e2:.2e}")
    assert e1 < 2e-6 and e2 < 2e-6
    out.update(attn_wseed=wseed, attn_xseed=xseed, attn_shape=np.array([2, N, dim, heads]), attn_scale=scale.view(-1).numpy(),
               attn_out_alibi=ref_b.numpy(), attn_out_plain=ref_n.numpy())
    np.savez_compressed(os.path.join(GOLD, "emotion2vec_parts.npz"), **out)




def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    ctx = import_reference()
    case_interp()
    case_nms()
    case_nms_modes()
    case_datasets()
    case_metrics()
    case_eval_detection()
    case_lr_schedules()
    case_byola_net()
    case_emotion2vec_parts()
    case_optimizer_split(ctx)
    case_tinydataset(ctx)
    case_forward("fwd_audio_exp10", R.make_cfg(**AUDIO_EXP10), [768, 601], 81, 82, ctx)
    case_train("train_audio_exp10", R.make_cfg(**AUDIO_EXP10), [768, 700, 768, 650], 81, 83, ctx)
    case_train_f64("train_audio_exp10_f64", "train_audio_exp10", R.make_cfg(**AUDIO_EXP10))
    case_forward("fwd_exp5", R.make_cfg(**EXP5), [768, 655], 91, 92, ctx)
    case_train("train_exp5", R.make_cfg(**EXP5), [768, 700, 640], 91, 93, ctx)
    case_train_f64("train_exp5_f64", "train_exp5", R.make_cfg(**EXP5))
    case_forward("fwd_small", R.make_cfg(**SMALL), [192, 150], 3, 4, ctx)
    case_forward("fwd_small_init", R.make_cfg(**SMALL), [192, 101], 5, 6, ctx, trained_like=False)
    case_forward("fwd_mid", R.make_cfg(**MID), [384, 300, 97], 7, 8, ctx)
    case_train("train_small", R.make_cfg(**SMALL), [192, 150, 180, 192], 3, 9, ctx)
    case_train("train_mid", R.make_cfg(**MID), [384, 300, 350, 200], 7, 10, ctx)
    case_forward("fwd_exp10", R.make_cfg(), [768, 700], 21, 22, ctx)
    case_train("train_exp10", R.make_cfg(), [768, 700, 768, 650], 21, 23, ctx)          # full-size exp10, 4 clips
    case_train_f64("train_exp10_f64", "train_exp10", R.make_cfg())
    case_forward("fwd_the_small", R.make_cfg(variant="the", **SMALL), [192, 133], 31, 32, ctx)
    case_train("train_the_small", R.make_cfg(variant="the", **SMALL), [192, 150, 180, 192], 33, 34, ctx)
    case_forward("fwd_plain_small", R.make_cfg(variant="plain", **SMALL), [192, 151], 51, 52, ctx)
    case_train("train_plain_small", R.make_cfg(variant="plain", **SMALL), [192, 150, 180, 192], 53, 54, ctx)
    case_forward("fwd_conv_small", R.make_cfg(variant="plainconv", **SMALL), [192, 133], 61, 62, ctx)
    case_train("train_conv_small", R.make_cfg(variant="plainconv", **SMALL), [192, 150, 180, 192], 63, 64, ctx)
    case_local_attention_tables()
    case_forward("fwd_relpe_small", R.make_cfg(use_rel_pe=True, **SMALL), [192, 141], 71, 72, ctx)
    case_train("train_relpe_small", R.make_cfg(use_rel_pe=True, **SMALL), [192, 150, 180, 192], 73, 74, ctx)
    case_forward("fwd_recon_small", R.make_cfg(variant="recon", **SMALL), [192, 140], 41, 42, ctx)
    case_train("train_recon_small", R.make_cfg(variant="recon", **SMALL), [192, 150, 180, 192], 43, 44, ctx)
    print("done")


def case_eval_detection():
    """The deepfake evaluator (libs/utils/Evaluation/eval.py:119-165, eval_detection.py:18-319): `run_evaluation` of the
    reference on synthetic validation results -> its prediction json and its AP at tIoU {0.5, 0.75, 0.9, 0.95}.  `np.float`
    (gone since numpy 1.24, eval_detection.py:310-311) is aliased HERE, in the generator process only; distinct scores, so
    the hash-order of the reference's video set cannot change the result."""
    import contextlib
    import io
    import json
    import tempfile
    if not hasattr(np, "float"):
        np.float = float
    from libs.utils.Evaluation import run_evaluation
    out = {}
    for ci, seed in enumerate((31, 32)):
        rs = np.random.RandomState(seed)
        gt_items, vids, ts, te, sc = [], [], [], [], []
        for v in range(14):
            vid = f"id{v:03d}/clip.mp4"
            n = int(rs.randint(0, 3)) if v else 2
            segs = [[float(s), float(s + rs.uniform(0.2, 1.5))] for s in rs.uniform(0, 8, n)]
            gt_items.append({"video_id": vid, "n_fakes": n, "segments_time": segs})
            if v % 6 == 5:
                continue                                   # no predictions for this video
            for (a, b) in segs:
                for _ in range(int(rs.randint(1, 4))):
                    j = rs.uniform(-0.25, 0.25, 2)
                    vids.append(vid); ts.append(a + j[0]); te.append(max(a + j[0] + 0.05, b + j[1])); sc.append(float(rs.uniform(0, 1)))
            for _ in range(int(rs.randint(0, 3))):
                s0 = float(rs.uniform(-0.3, 9))
                vids.append(vid); ts.append(s0); te.append(s0 + float(rs.uniform(0.1, 1))); sc.append(float(rs.uniform(0, 1)))
        assert len(set(sc)) == len(sc)
        preds = {"video-id": vids, "t-start": np.asarray(ts, np.float32), "t-end": np.asarray(te, np.float32),
                 "label": np.zeros(len(vids), np.int64), "score": np.asarray(sc, np.float32)}
        cls_file = None
        if ci == 1:                                        # with the external video-level scores
            f = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False)
            json.dump({g["video_id"]: [float(rs.uniform(0.2, 1)), float(rs.uniform(0.2, 1))] for g in gt_items}, f)
            f.close()
            cls_file = f.name
        pf = tempfile.NamedTemporaryFile("w", suffix=".json", delete=False).name
        # the reference fans the per-video formatting out to 16 loky PROCESSES (eval.py:137, a soft `prefer=`); fresh
        # worker processes cannot re-import `libs` here (the torchsort stub lives in this process only), so the same
        # joblib call is served by threads -- same function, same results
        import joblib
        with contextlib.redirect_stdout(io.StringIO()), joblib.parallel_backend("threading"):
            mAP, _ = run_evaluation(preds, gt_items, pf, cls_score_file=cls_file)
        from libs.utils.Evaluation.eval_detection import ANETdetection
        with contextlib.redirect_stdout(io.StringIO()):
            det = ANETdetection(ground_truth_filename=gt_items, prediction_filename=pf, subset="test")
            m, avg = det.evaluate()
        pj = json.load(open(pf))
        out[f"c{ci}_seed"] = np.array(seed)
        out[f"c{ci}_gt"] = np.array(json.dumps(gt_items))
        out[f"c{ci}_cls"] = np.array(json.dumps(json.load(open(cls_file))) if cls_file else "")
        out[f"c{ci}_vid"] = np.array(vids)
        out[f"c{ci}_tab"] = np.stack([preds["t-start"], preds["t-end"], preds["score"]], 1)
        out[f"c{ci}_mAP"], out[f"c{ci}_avg"], out[f"c{ci}_ret"] = m, np.array(avg), np.array(mAP)
        out[f"c{ci}_pred_json"] = np.array(json.dumps(pj, sort_keys=True))
        out[f"c{ci}_report"] = np.array(open(pf.replace(".json", ".txt")).read())
        print(f"[eval_detection] case {ci}: mAP {m}, mean x100 {mAP:.3f}")
    np.savez_compressed(os.path.join(GOLD, "eval_detection.npz"), **out)


def case_train_f64(name, base, cfg, ctx=None):
    """float64 ground truth for the sampled gradients of a full-size training fixture: the oracle -- pinned to the reference
    in fp32 by `base` -- re-run in float64 on the same weights / inputs / targets.  Stores the fp64 sample of every gradient of
    `base` next to the error of the REFERENCE's own fp32 gradient against it, i.e. what fp32 rounding alone costs."""
    g = np.load(os.path.join(GOLD, f"{base}.npz"), allow_pickle=True)
    lens = [int(v) for v in g["lens"]]
    x, mask = synth_inputs(cfg, lens, int(g["xseed"]))
    sd = R.synth_state_dict(R.param_shapes(cfg), int(g["wseed"]), True)
    sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    gts = [torch.from_numpy(np.asarray(s_, np.float32)).double() if f else None for s_, f in zip(g["gt_segments"], g["is_fake"])]
    gls = [None if s_ is None else torch.zeros(len(s_), dtype=torch.int64) for s_ in gts]
    out = R.model_forward(sd64, cfg, x.double(), mask.double(), training=True)
    ol, norm, _ = R.losses(cfg, out, gts, gls, cfg["init_loss_norm"], x_in=x.double())
    ol["final_loss"].backward()
    store = {"base": np.array(base), "final_loss": np.array(float(ol["final_loss"]))}
    print(f"[{name}] fp64 final_loss {float(ol['final_loss']):.9f} (fp32 reference: {float(g['losses'][3]):.9f})")
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        flat = sd64[key[6:]].grad.flatten()
        stride = max(1, flat.numel() // 4096)
        ref64 = flat[::stride][:4096].numpy().copy()
        ref32 = g[key].astype(np.float64)
        e = float(np.abs(ref32 - ref64).max() / (np.abs(ref64).max() + 1e-300))
        store["grad64::" + key[6:]] = ref64
        store["referr::" + key[6:]] = np.array(e)
        print(f"[{name}] {key[6:]}: reference fp32 vs fp64 ground truth, max rel. element error {e:.3e}")
    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **store)


def case_local_attention_tables():
    """The reference's LocalMaskedMHCA (blocks.py:883-1224) built with use_rel_pe=True AND use_time_weight=True on seeded
    weights / inputs: output and the gradients of the input and of both tables.  (use_time_weight is reachable only by
    constructing blocks directly -- the meta-archs pass use_rel_pe alone -- so it is pinned at the operator level.)"""
    from libs.modeling.blocks import LocalMaskedMHCA
    B, C, T, H, W = 2, 32, 48, 4, 7
    rs = np.random.RandomState(9001)
    m = LocalMaskedMHCA(C, H, window_size=W, n_qx_stride=1, n_kv_stride=1, use_rel_pe=True, use_time_weight=True)
    sd = {}
    for k, v in m.state_dict().items():
        if k.endswith("norm.weight") or k == "time_weighting":
            sd[k] = torch.from_numpy((1.0 + 0.2 * rs.standard_normal(v.shape)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((0.3 * rs.standard_normal(v.shape)).astype(np.float32))
    m.load_state_dict(sd)
    x = torch.from_numpy(rs.standard_normal((B, C, T)).astype(np.float32)).requires_grad_(True)
    mask = torch.ones(B, 1, T, dtype=torch.bool)
    mask[1, 0, 30:] = False
    out, _ = m(x, mask)
    w = torch.from_numpy(rs.standard_normal(out.shape).astype(np.float32))
    (out * w).sum().backward()
    # the oracle on the same weights
    xo = x.detach().clone().requires_grad_(True)
    sdo = {"attn." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oo, _ = R.mhca(sdo, "attn.", xo, mask.float(), xo, mask.float(), xo, mask.float(), H, W, 1)
    (oo * w).sum().backward()
    named = dict(m.named_parameters())
    errs = {"out": maxdiff(out, oo), "dx": maxdiff(x.grad, xo.grad), "drel": maxdiff(named["rel_pe"].grad, sdo["attn.rel_pe"].grad),
            "dtw": maxdiff(named["time_weighting"].grad, sdo["attn.time_weighting"].grad)}
    print("[local_attention_tables] oracle vs reference:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert all(v < 2e-5 for v in errs.values())
    np.savez_compressed(os.path.join(GOLD, "local_attention_tables.npz"), seed=9001, dims=np.array([B, C, T, H, W]),
                        out=out.detach().numpy(), dx=x.grad.numpy(), drel=named["rel_pe"].grad.numpy(),
                        dtw=named["time_weighting"].grad.numpy(), dqw=named["query.weight"].grad.numpy())


NAMED = {
    "metrics": lambda ctx: case_metrics(),
    "nms_modes": lambda ctx: case_nms_modes(),
    "lr": lambda ctx: case_lr_schedules(),
    "eval_detection": lambda ctx: case_eval_detection(),
    "rel_pe": lambda ctx: (case_local_attention_tables(),
                           case_forward("fwd_relpe_small", R.make_cfg(use_rel_pe=True, **SMALL), [192, 141], 71, 72, ctx),
                           case_train("train_relpe_small", R.make_cfg(use_rel_pe=True, **SMALL), [192, 150, 180, 192], 73, 74, ctx)),
    "train_f64": lambda ctx: (case_train_f64("train_exp10_f64", "train_exp10", R.make_cfg()),
                              case_train_f64("train_audio_exp10_f64", "train_audio_exp10", R.make_cfg(**AUDIO_EXP10)),
                              case_train_f64("train_exp5_f64", "train_exp5", R.make_cfg(**EXP5))),
    "byola_net": lambda ctx: case_byola_net(),
    "emotion2vec_parts": lambda ctx: case_emotion2vec_parts(),
    "optimizer_split": case_optimizer_split,
    "tinydataset": case_tinydataset,
    "train_exp10": lambda ctx: case_train("train_exp10", R.make_cfg(), [768, 700, 768, 650], 21, 23, ctx),
    "audio_exp10": lambda ctx: (case_forward("fwd_audio_exp10", R.make_cfg(**AUDIO_EXP10), [768, 601], 81, 82, ctx),
                                case_train("train_audio_exp10", R.make_cfg(**AUDIO_EXP10), [768, 700, 768, 650], 81, 83, ctx)),
    "exp5": lambda ctx: (case_forward("fwd_exp5", R.make_cfg(**EXP5), [768, 655], 91, 92, ctx),
                         case_train("train_exp5", R.make_cfg(**EXP5), [768, 700, 640], 91, 93, ctx)),
    "plainconv": lambda ctx: (case_forward("fwd_conv_small", R.make_cfg(variant="plainconv", **SMALL), [192, 133], 61, 62, ctx),
                              case_train("train_conv_small", R.make_cfg(variant="plainconv", **SMALL), [192, 150, 180, 192], 63, 64, ctx)),
    "plain": lambda ctx: (case_forward("fwd_plain_small", R.make_cfg(variant="plain", **SMALL), [192, 151], 51, 52, ctx),
                          case_train("train_plain_small", R.make_cfg(variant="plain", **SMALL), [192, 150, 180, 192], 53, 54, ctx)),
}


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1:                        # regenerate the named fixtures only
        ctx = import_reference()
        for nm in sys.argv[1:]:
            NAMED[nm](ctx)
    else:
        main()
