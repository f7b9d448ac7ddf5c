"""BASELINE config 5 in miniature: raw PCM + uint8 frames -> the three HIP encoders -> K1 -> UMMAFormer -> decode / NMS, against
the SAME chain built from the CPU oracles (oracle/c3d_ref.py, byola_ref.py, emotion2vec_ref.py, ummaformer_ref.py).  The
localisation model and NMS oracles are pinned to the reference; the encoder oracles are restatements,
so this test pins the plumbing between the stages -- trims, stream order, interpolation, metadata -- and the end-to-end numerics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_raw_media_to_segments_matches_chained_oracles():
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.extractors import (LogMelFrontEnd, AudioNTT2020Task6, Emotion2Vec, C3DVideoEncoder,
                                                                 byola_features, emotion2vec_features, video_features)
    from audio_visual_deepfake_detection_amd.extractors import c3d as C
    from oracle import byola_ref as BR, c3d_ref as CR, emotion2vec_ref as ER, ummaformer_ref as R
    dev = torch.device("cuda:0")
    torch.manual_seed(23)
    fe, byola = LogMelFrontEnd(), AudioNTT2020Task6().eval()
    e2v, c3d = Emotion2Vec(depth=2, prenet_depth=2), C3DVideoEncoder()
    with torch.no_grad():
        e2v.modality_encoders["AUDIO"].alibi_scale.uniform_(0.5, 1.5)
        for i in (1, 5, 9):
            byola.features[i].running_mean.normal_(0, 0.3)
            byola.features[i].running_var.uniform_(0.5, 1.5)
    sd_b = {k: v.clone() for k, v in byola.state_dict().items()}
    sd_e = {k: v.clone() for k, v in e2v.state_dict().items()}
    sd_c = {k: v.clone() for k, v in c3d.state_dict().items()}
    n_frames, fps = 40, 25.0
    duration = n_frames / fps                                        # 1.6 s of media
    n_pcm = int(duration * 16000)
    g = torch.Generator().manual_seed(77)
    pcm = torch.rand(n_pcm, generator=g) - 0.5
    frames = torch.randint(0, 256, (n_frames, 72, 64, 3), generator=g, dtype=torch.uint8)

    # ---- product chain on the MI355X ---------------------------------------------------------------------------
    fe, byola, e2v, c3d = fe.to(dev), byola.to(dev), e2v.to(dev), c3d.to(dev)
    old = C.CHUNK
    C.CHUNK = 16                                                     # 40 frames -> 3 chunks (16, 16, 8 kept): stitch logic on
    try:
        v = video_features(frames.to(dev), c3d)
    finally:
        C.CHUNK = old
    a1 = byola_features(pcm.to(dev), byola, fe)
    a2 = emotion2vec_features(pcm.to(dev), e2v)
    n1, n2 = R.trim_lengths(duration)                               # dataset-side trim, deepfake_video_audio.py:1683-1684
    streams = [v.cpu().numpy(), a1[:n1].cpu().numpy(), a2[:n2].cpu().numpy()]
    assert streams[0].shape == (40, 256) and streams[1].shape == (n1, 2048) and streams[2].shape == (n2, 768)
    cfg = R.make_cfg(input_dim=3072, embd_dim=32, fpn_dim=32, head_dim=32, n_head=4)
    sd_m = R.synth_state_dict(R.param_shapes(cfg), 37, True)
    model, _ = build_model("deepfake_exp10.yaml", embd_dim=32, fpn_dim=32, head_dim=32, n_head=4)
    model.load_state_dict(sd_m)
    model = model.to(dev).eval()
    feat_stride = float(n_frames) / 768
    meta = dict(fps=fps, duration=duration, feat_stride=feat_stride, feat_num_frames=feat_stride)
    res = model([dict(video_id="raw.mp4", streams=streams, crop=None, **meta)])[0]

    # ---- the same chain from the CPU oracles ---------------------------------------------------------------------
    torch.set_num_threads(8)
    ov = CR.video_features(sd_c, frames, chunk=16)
    o1 = BR.byola_forward(sd_b, BR.log_mel(pcm)[None, None])[0]
    o2 = ER.forward(sd_e, pcm[None], heads=12, prenet_depth=2, depth=2)[0]
    for name, got, ref in (("video", v, ov), ("byol-a", a1, o1), ("emotion2vec", a2, o2)):
        assert got.shape == ref.shape, name
        assert float((got.cpu() - ref).abs().max() / ref.abs().max()) < 1e-3, name
    feats = R.interp_concat([ov.numpy(), o1[:n1].numpy(), o2[:n2].numpy()], 768)
    o = R.infer_clip(sd_m, cfg, feats, meta)
    # logits first (the north star's 1e-3), then the detections
    x, mask = model.preprocessing([dict(video_id="raw.mp4", streams=streams, crop=None, **meta)])
    with torch.no_grad():
        logits, offsets, _, cls_scores, _ = model.forward_features(x, mask)
    assert float((logits[0].cpu() - o["logits"][:, 0]).abs().max()) < 1e-3
    assert float((offsets[0].cpu() - o["offsets"]).abs().max()) < 1e-3
    k = o["segments"].shape[0]
    assert res["segments"].shape[0] == k
    assert sorted(res["kept_points"].tolist()) == sorted(o["kept_points"].tolist())
    order = {int(p): i for i, p in enumerate(res["kept_points"].tolist())}
    perm = torch.tensor([order[int(p)] for p in o["kept_points"].tolist()], dtype=torch.int64)
    assert float((res["segments"][perm] - o["segments"]).abs().max()) < 1e-3
    assert float((res["scores"][perm] - o["scores"]).abs().max()) < 1e-4
    assert float((res["video_cls"] - o["video_cls"]).abs().max()) < 1e-3


def test_config5_full_size_clip_through_batched_extractors_and_exp10_model():
    """BASELINE config 5 at FULL size (round-3 verdict item 5): one 768-frame 224x224 clip + 491 520 PCM samples through
    `extract_clip_batch` -> K1 -> the full-size exp10 model -> decode + NMS.  No CPU oracle runs here (C3D alone is
    3.5 TFLOP per clip: minutes on the host); what is held is the reference's output shapes
    (extract_video_features.py:53-77 -> [768, 256]; scripts/extract_audio_feature_batch.py:73-98 -> [383, 2048] after the
    dataset trim and [1535, 768]), finiteness, run-to-run identical detections, and equality of the batched extraction
    entry with the per-stage HIP functions the miniature test pins to the oracles."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.extractors import (LogMelFrontEnd, AudioNTT2020Task6, Emotion2Vec, C3DVideoEncoder,
                                                                 byola_features, emotion2vec_features, video_features,
                                                                 extract_clip_batch)
    from oracle import ummaformer_ref as R
    dev = torch.device("cuda:0")
    torch.manual_seed(29)
    fe, byola = LogMelFrontEnd().to(dev), AudioNTT2020Task6().to(dev).eval()
    e2v, c3d = Emotion2Vec().to(dev), C3DVideoEncoder().to(dev)          # default depth: 4 prenet + 8 blocks
    n_frames, n_pcm, fps = 768, 491520, 25.0
    duration = n_frames / fps
    g = torch.Generator(device=dev).manual_seed(3001)
    pcm = torch.rand((1, n_pcm), generator=g, device=dev) - 0.5
    frames = torch.randint(0, 256, (1, n_frames, 224, 224, 3), generator=g, device=dev, dtype=torch.uint8)
    n1, n2 = R.trim_lengths(duration)
    assert (n1, n2) == (383, 1535)
    outs = [(torch.zeros((768, 256), device=dev), torch.zeros((n1, 2048), device=dev), torch.zeros((n2, 768), device=dev))]
    extract_clip_batch(frames, pcm, c3d, byola, fe, e2v, outs, torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    v, a1, a2 = outs[0]
    # per-stage entry points on the same media (raw outputs before the dataset-side trim)
    pv = video_features(frames[0], c3d)
    p1 = byola_features(pcm[0], byola, fe)
    p2 = emotion2vec_features(pcm[0], e2v)
    assert tuple(pv.shape) == (768, 256) and p1.shape[1] == 2048 and p1.shape[0] >= n1 and tuple(p2.shape) == (n2, 768)
    assert torch.equal(v, pv) and torch.equal(a1, p1[:n1])
    # the batched Emotion2Vec pass stacks clips as GEMM rows: same kernels, same per-row arithmetic
    assert float((a2 - p2).abs().max()) <= 1e-5 * float(p2.abs().max())
    for t in (v, a1, a2):
        assert torch.isfinite(t).all() and float(t.abs().max()) > 0

    torch.manual_seed(31)
    model, _ = build_model("deepfake_exp10.yaml")
    model = model.to(dev).eval()
    meta = dict(fps=fps, duration=duration, feat_stride=1.0, feat_num_frames=1.0)
    clip = dict(video_id="full.mp4", streams=[v.cpu().numpy(), a1.cpu().numpy(), a2.cpu().numpy()], crop=None, **meta)
    x, mask = model.preprocessing([clip])
    assert tuple(x.shape) == (1, 768, 3072) and float(mask.sum()) == 768
    res = [model([clip])[0] for _ in range(2)]
    for r in res:
        assert r["segments"].shape[0] == r["scores"].shape[0] <= 100 and r["segments"].shape[0] > 0
        assert torch.isfinite(r["segments"]).all() and torch.isfinite(r["scores"]).all()
        assert float(r["segments"].min()) >= 0.0 and float(r["segments"].max()) <= duration
    assert res[0]["kept_points"].tolist() == res[1]["kept_points"].tolist()
    assert torch.equal(res[0]["segments"], res[1]["segments"]) and torch.equal(res[0]["scores"], res[1]["scores"])


def test_emotion2vec_default_depth_matches_restatement_on_short_clip():
    """Default-depth Emotion2Vec (4 prenet + 8 blocks, 12 heads, 768 wide: upstream/models/emotion2vec.py config) against the
    torch restatement on a 2 s clip (the miniature e2e test runs depth 2 + 2 only).  Tolerance 1e-3 of the output scale."""
    from audio_visual_deepfake_detection_amd.extractors import Emotion2Vec, emotion2vec_features
    from oracle import emotion2vec_ref as ER
    dev = torch.device("cuda:0")
    torch.manual_seed(41)
    e2v = Emotion2Vec()
    with torch.no_grad():
        e2v.modality_encoders["AUDIO"].alibi_scale.uniform_(0.5, 1.5)
    sd = {k: v.clone() for k, v in e2v.state_dict().items()}
    g = torch.Generator().manual_seed(43)
    pcm = torch.rand(32000, generator=g) - 0.5
    got = emotion2vec_features(pcm.to(dev), e2v.to(dev)).cpu()
    torch.set_num_threads(8)
    ref = ER.forward(sd, pcm[None])[0][:got.shape[0]]
    assert got.shape == (int(50 * 2.0 - 0.817), 768) == tuple(ref.shape)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-3
