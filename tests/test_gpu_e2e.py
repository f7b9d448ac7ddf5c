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
