"""BYOL-A extraction path on a real MI355X vs the torch CPU restatement (oracle/byola_ref.py).
The network is pinned to the reference's own class (tests/golden/byola_net.npz); the log-mel front-end is PARITY UNPINNED
upstream (no torchaudio in the reference tree or this image): for it this pins HIP == torch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(seed=0):
    from audio_visual_deepfake_detection_amd.extractors import AudioNTT2020Task6
    torch.manual_seed(seed)
    m = AudioNTT2020Task6(64, 2048)
    with torch.no_grad():          # non-trivial BatchNorm statistics
        for i in (1, 5, 9):
            m.features[i].running_mean.normal_(0, 0.3)
            m.features[i].running_var.uniform_(0.5, 1.5)
            m.features[i].weight.normal_(1.0, 0.1)
            m.features[i].bias.normal_(0, 0.1)
    return m.eval()


@pytest.mark.parametrize("n", [16000, 48000 + 37, 160 * 200])
def test_log_mel_front_end(n):
    from audio_visual_deepfake_detection_amd.extractors import LogMelFrontEnd
    from oracle import byola_ref as R
    g = torch.Generator().manual_seed(n)
    pcm = (torch.rand(n, generator=g) - 0.5)
    fe = LogMelFrontEnd().to("cuda:0")
    got = fe(pcm.to("cuda:0")).cpu()
    ref = R.log_mel(pcm)
    assert got.shape == ref.shape == (64, 1 + n // 160)
    err = float((got - ref).abs().max())
    print(f"log-mel n={n}: max abs err {err:.3e}")
    assert err < 1e-3


def test_byola_network_matches_reference_golden(golden_dir):
    """HIP BYOL-A network == the reference's AudioNTT2020Task6 output stored in tests/golden/byola_net.npz."""
    import os
    from audio_visual_deepfake_detection_amd.extractors import AudioNTT2020Task6
    from oracle import byola_ref as R
    g = np.load(os.path.join(golden_dir, "byola_net.npz"))
    m = AudioNTT2020Task6(64, 2048)
    m.load_state_dict(R.seeded_state_dict(int(g["wseed"])), strict=True)
    lms = torch.from_numpy(np.random.RandomState(int(g["xseed"])).randn(*g["lms_shape"]).astype(np.float32))
    got = m.eval().to("cuda:0")(lms.to("cuda:0")).cpu()
    ref = torch.from_numpy(g["out"])
    assert got.shape == ref.shape
    rel = float((got - ref).abs().max() / ref.abs().max())
    print(f"BYOL-A net vs reference golden: rel err {rel:.3e}")
    assert rel < 1e-4


def test_byola_network_and_pipeline():
    from audio_visual_deepfake_detection_amd.extractors import LogMelFrontEnd, byola_features
    from oracle import byola_ref as R
    m = _model()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    lms = torch.randn(2, 1, 64, 203, generator=g)
    ref = R.byola_forward(sd, lms)
    md = m.to("cuda:0")
    got = md(lms.to("cuda:0")).cpu()
    assert got.shape == ref.shape == (2, 203 // 8, 2048)
    rel = float((got - ref).abs().max() / ref.abs().max())
    print(f"BYOL-A net: rel err {rel:.3e}")
    assert rel < 1e-4
    # PCM -> features, the shape contract of the saved .npy (768-frame clip: 491520 samples -> [384, 2048])
    pcm = torch.rand(491520, generator=g) - 0.5
    fe = LogMelFrontEnd().to("cuda:0")
    feats = byola_features(pcm.to("cuda:0"), md, fe).cpu()
    ref2 = R.byola_forward(sd, R.log_mel(pcm)[None, None])[0]
    assert feats.shape == ref2.shape == (384, 2048)
    rel2 = float((feats - ref2).abs().max() / ref2.abs().max())
    print(f"BYOL-A pipeline: rel err {rel2:.3e}")
    assert rel2 < 1e-3


def test_reference_checkpoint_key_filter():
    m = _model()
    sd = {"model.features." + k[len("features."):] if k.startswith("features.") else "net.fc." + k[len("fc."):]: v
          for k, v in m.state_dict().items()}
    m2 = _model(seed=1)
    m2.load_weight(None, "cpu", state_dict={"state_dict": sd})
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


@pytest.mark.parametrize("n,per_layer", [(16000 * 2 + 123, True), (9000, False)])
def test_emotion2vec_matches_torch_restatement(n, per_layer):
    from audio_visual_deepfake_detection_amd.extractors import Emotion2Vec, emotion2vec_features
    from oracle import emotion2vec_ref as R
    torch.manual_seed(3)
    m = Emotion2Vec(depth=2, prenet_depth=2, learned_alibi_scale_per_layer=per_layer)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("alibi_scale"):
                p.uniform_(0.5, 1.5)
            elif p.dim() == 1 and "norm" in name or ".2.1." in name or "project_features.1" in name:
                if name.endswith("weight"):
                    p.normal_(1.0, 0.1)
                else:
                    p.normal_(0.0, 0.1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(n)
    pcm = torch.rand(n, generator=g) - 0.5
    ref = R.forward(sd, pcm[None], heads=12, prenet_depth=2, depth=2)[0]
    md = m.to("cuda:0")
    got = md.forward_clip(pcm.to("cuda:0")).cpu()
    assert got.shape == ref.shape == ((n - 400) // 320 + 1, 768)
    rel = float((got - ref).abs().max() / ref.abs().max())
    print(f"emotion2vec n={n}: rel err {rel:.3e}")
    assert rel < 1e-3
    trimmed = emotion2vec_features(pcm.to("cuda:0"), md)
    assert trimmed.shape[0] == int(50 * (n / 16000.0) - 0.817)


def _c3d_model(batch_norm=False, feats=(64, 96, 128, 128)):
    from audio_visual_deepfake_detection_amd.extractors import C3DVideoEncoder
    torch.manual_seed(11)
    m = C3DVideoEncoder(n_features=feats, batch_norm=batch_norm)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if ".bn." in name:
                p.normal_(1.0 if name.endswith("weight") else 0.0, 0.1)
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.normal_(0.0, 0.1)
            elif name.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    return m


@pytest.mark.parametrize("cin,cout,T,H,W", [(4, 64, 3, 16, 40), (64, 96, 2, 32, 32), (96, 128, 3, 16, 16),
                                            (128, 128, 5, 8, 8), (128, 128, 6, 4, 4), (64, 64, 2, 19, 35),
                                            (8, 32, 3, 5, 7)])
@pytest.mark.parametrize("mode", [0, 1])
def test_conv3d_k3_matches_torch(cin, cout, T, H, W, mode):
    """mode 0: exact fp32 MFMA; mode 1: split-bf16 operands (layers with Cin >= 16; others keep the exact kernel)."""
    from audio_visual_deepfake_detection_amd._lib import lib, check
    g = torch.Generator().manual_seed(cin * 1000 + cout + H)
    B = 2
    x = torch.randn(B, cin, T, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    b = torch.randn(cout, generator=g)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv3d(x.double(), w.double(), b.double(), padding=1), 0.01)
    xd = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    wp = w.permute(2, 3, 4, 0, 1).contiguous().cuda()
    y = torch.empty(B, T, H, W, cout, device="cuda")
    check(lib.avdf_conv3d_mode(mode), "mode")
    try:
        check(lib.avdf_conv3d_k3(xd.data_ptr(), wp.data_ptr(), b.cuda().data_ptr(), y.data_ptr(), B, T, H, W, cin, cout, 3,
                                 torch.cuda.current_stream().cuda_stream), "conv3d")
    finally:
        check(lib.avdf_conv3d_mode(0), "mode")
    got = y.permute(0, 4, 1, 2, 3).cpu().double()
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"conv3d {cin}->{cout} {T}x{H}x{W} mode {mode}: rel err {err:.2e}")
    assert err < (2e-6 if mode == 0 else 2e-5)


def test_video_prep_and_pool_bit_exact():
    from audio_visual_deepfake_detection_amd._lib import lib, check
    from oracle import c3d_ref as R
    g = torch.Generator().manual_seed(5)
    fr = torch.randint(0, 256, (5, 224, 200, 3), generator=g, dtype=torch.uint8)
    ref = R.prepare_chunks(fr, chunk=8)[0]                      # [3, 8, 96, 96]
    out = torch.empty(8, 96, 96, 4, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    check(lib.avdf_video_prep(fr.cuda().data_ptr(), out.data_ptr(), 5, 224, 200, 8, 96, st), "prep")
    assert torch.equal(out[..., :3].cpu(), ref.permute(1, 2, 3, 0)) and float(out[..., 3].abs().max()) == 0.0
    x = torch.randn(4, 12, 18, 8, generator=g)
    for p in (2, 3):
        y = torch.empty(4, 12 // p, 18 // p, 8, device="cuda")
        check(lib.avdf_maxpool_hw(x.cuda().data_ptr(), y.data_ptr(), 4, 12, 18, 8, p, st), "pool")
        r = torch.nn.functional.max_pool3d(x.permute(3, 0, 1, 2)[None], (1, p, p))[0].permute(1, 2, 3, 0)
        assert torch.equal(y.cpu(), r)


@pytest.mark.parametrize("batch_norm", [False, True])
def test_c3d_encoder_matches_torch_restatement(batch_norm):
    from oracle import c3d_ref as R
    m = _c3d_model(batch_norm)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(2)
    video = torch.rand(1, 3, 4, 96, 96, generator=g)
    ref = R.encoder_forward(sd, video)
    got = m.cuda()(video.cuda()).cpu()
    assert got.shape == ref.shape == (1, 256, 4)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"C3D encoder bn={batch_norm}: rel err {err:.2e}")
    assert err < 1e-4


def test_video_features_chunking_and_tail_trim():
    """chunk/pad/stitch rule on a small chunk size (the stitch logic is size-independent) + exact tail trimming."""
    from audio_visual_deepfake_detection_amd.extractors import c3d as C
    from oracle import c3d_ref as R
    m = _c3d_model(False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda()
    g = torch.Generator().manual_seed(9)
    fr = torch.randint(0, 256, (21, 120, 100, 3), generator=g, dtype=torch.uint8)
    old = C.CHUNK
    C.CHUNK = 16
    try:
        assert [c[2] for c in C.chunk_plan(32, 16)] == [16, 16, 0]
        got_trim = C.video_features(fr.cuda(), m, trim_tail=True).cpu()
        got_full = C.video_features(fr.cuda(), m, trim_tail=False).cpu()
    finally:
        C.CHUNK = old
    ref = R.video_features(sd, fr, chunk=16)
    assert got_full.shape == ref.shape == (21, 256)
    assert torch.equal(got_trim, got_full)                       # trimming the all-zero tail is exact
    err = float((got_full - ref).abs().max() / ref.abs().max())
    print(f"video_features: rel err {err:.2e}")
    assert err < 1e-4


def test_extract_clip_batch_streams_equal_sequential():
    """The two-stream overlap of the audio and video encoders (extractors.extract_clip_batch) changes no value."""
    from audio_visual_deepfake_detection_amd.extractors import (LogMelFrontEnd, AudioNTT2020Task6, Emotion2Vec, C3DVideoEncoder,
                                                                 extract_clip_batch)
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    fe, byola = LogMelFrontEnd().to(dev), AudioNTT2020Task6().to(dev).eval()
    e2v, c3d = Emotion2Vec(depth=2, prenet_depth=1).to(dev), C3DVideoEncoder().to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    B, T = 3, 24
    pcm = torch.rand((B, 16000), generator=g, device=dev) - 0.5
    frames = torch.randint(0, 256, (B, T, 64, 64, 3), generator=g, device=dev, dtype=torch.uint8)
    mk = lambda: [(torch.zeros(T, 256, device=dev), torch.zeros(12, 2048, device=dev), torch.zeros(49, 768, device=dev))
                  for _ in range(B)]
    seq, par = mk(), mk()
    extract_clip_batch(frames, pcm, c3d, byola, fe, e2v, seq, None)
    torch.cuda.synchronize()
    for _ in range(2):                       # twice: the second pass reuses cached allocations across the two streams
        extract_clip_batch(frames, pcm, c3d, byola, fe, e2v, par, torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    for b in range(B):
        for a, c, name in zip(seq[b], par[b], ("video", "byol-a", "emotion2vec")):
            assert float(a.abs().max()) > 0, name
            assert torch.equal(a, c), f"clip {b} {name}: streamed extraction differs"


@pytest.mark.parametrize("H,T,ld", [(3, 27, 28), (12, 1535, 1536), (2, 700, 704), (2, 1100, 1100), (1, 2040, 2048), (1, 2100, 2112),
                                    (2, 50, 51)])
def test_softmax_alibi_heads_all_paths(H, T, ld):
    """softmax_j(S*scale - slope_h*|i-j|) in place, padding columns zeroed: the register-resident kernels (ld <= 2048,
    ld % 4 == 0: 2 / 4 / 6 / 8 float4 per lane) and the scalar fallback (odd ld, ld > 2048) against float64."""
    from audio_visual_deepfake_detection_amd._lib import lib, check
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(T)
    s = torch.randn(H, T, ld, generator=g) * 3.0
    slopes = torch.tensor([2.0 ** (-(h + 1)) for h in range(H)])
    qk = 0.125
    i = torch.arange(T)[:, None].double()
    j = torch.arange(T)[None, :].double()
    ref = torch.softmax(s[:, :, :T].double() * qk - slopes.double()[:, None, None] * (i - j).abs(), dim=-1)
    sd = s.to(dev).contiguous()
    sl = slopes.to(dev)
    check(lib.avdf_softmax_alibi_heads(sd.data_ptr(), H, T, ld, T * ld, qk, sl.data_ptr(), torch.cuda.current_stream().cuda_stream),
          "softmax_alibi_heads")
    out = sd.cpu().double()
    assert float((out[:, :, :T] - ref).abs().max()) < 2e-6
    assert float(out[:, :, T:].abs().max()) == 0.0 if ld > T else True


def test_byola_batch_pad_mode_equals_reference_collate_semantics():
    """extract_audio_feature_batch.py:65-81,111-123: normalised log-mels zero-padded to the batch maximum, network on the padded
    images, features saved untrimmed.  Checked against the (reference-pinned) network oracle run on the SAME padded batch; the
    shorter clip must differ from its un-padded features near its end (that is the point of the mode)."""
    from audio_visual_deepfake_detection_amd.extractors import LogMelFrontEnd, byola_features, byola_features_batch
    from oracle import byola_ref as R
    m = _model(5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    pcms = [torch.rand(n, generator=g) - 0.5 for n in (48000, 30000 + 77, 16000)]
    md, fe = m.to("cuda:0"), LogMelFrontEnd().to("cuda:0")
    got = byola_features_batch([p.to("cuda:0") for p in pcms], md, fe)
    lms = [R.log_mel(p) for p in pcms]
    tmax = max(l.shape[1] for l in lms)
    batch = torch.zeros(len(pcms), 1, 64, tmax)
    for i, l in enumerate(lms):
        batch[i, 0, :, :l.shape[1]] = l
    ref = R.byola_forward(sd, batch)
    for i in range(len(pcms)):
        assert got[i].shape == ref[i].shape == (tmax // 8, 2048)
        assert float((got[i].cpu() - ref[i]).abs().max() / ref[i].abs().max()) < 1e-3
    alone = byola_features(pcms[2].to("cuda:0"), md, fe)
    n = alone.shape[0]
    assert float((got[2][:n - 2] - alone[:n - 2]).abs().max()) < 1e-3 * float(alone.abs().max())     # interior rows agree
    assert float((got[2][n - 1] - alone[n - 1]).abs().max()) > 1e-3 * float(alone.abs().max())       # the boundary row does not


def test_emotion2vec_batch_pad_mode_equals_reference_collate_semantics():
    """scripts/extract_audio_feature_batch.py:35-52,73-91: zero-padded PCM, padding_mask=None (attention sees the padding),
    rows [: int(50 * len - 0.817)] per clip -- against the restatement run on the same zero-padded batch."""
    from audio_visual_deepfake_detection_amd.extractors import Emotion2Vec, emotion2vec_features, emotion2vec_features_batch
    from oracle import emotion2vec_ref as R
    torch.manual_seed(13)
    m = Emotion2Vec(depth=1, prenet_depth=1)
    with torch.no_grad():
        m.modality_encoders["AUDIO"].alibi_scale.uniform_(0.5, 1.5)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(31)
    pcms = [torch.rand(n, generator=g) - 0.5 for n in (24000, 16000 + 50)]
    md = m.to("cuda:0")
    got = emotion2vec_features_batch([p.to("cuda:0") for p in pcms], md)
    batch = torch.zeros(2, 24000)
    for i, p in enumerate(pcms):
        batch[i, :p.numel()] = p
    ref = R.forward(sd, batch, heads=12, prenet_depth=1, depth=1)
    for i, p in enumerate(pcms):
        k = int(50 * (p.numel() / 16000.0) - 0.817)
        assert got[i].shape == (k, 768)
        assert float((got[i].cpu() - ref[i, :k]).abs().max() / ref[i].abs().max()) < 1e-3
    alone = emotion2vec_features(pcms[1].to("cuda:0"), md)
    assert float((got[1] - alone).abs().max()) > 1e-3 * float(alone.abs().max())      # the padded clip is "contaminated"


def test_emotion2vec_attention_matches_reference_alt_attention(golden_dir):
    """The HIP attention of one layer (q/k/v GEMMs, QK^T + ALiBi + softmax, PV, proj) against the output of the REFERENCE's own
    AltAttention + get_alibi_bias (tests/golden/emotion2vec_parts.npz, written by oracle/make_golden.py emotion2vec_parts)."""
    import os
    from audio_visual_deepfake_detection_amd.extractors import Emotion2Vec
    from tests.test_extractors_cpu import attn_fixture_inputs
    g = np.load(os.path.join(golden_dir, "emotion2vec_parts.npz"))
    sd, x, scale = attn_fixture_inputs(g)
    B, N, dim, heads = [int(v) for v in g["attn_shape"]]
    m = Emotion2Vec(depth=1, prenet_depth=1)
    blk = m.blocks[0]
    blk.attn.load_state_dict({k[len("attn."):]: v for k, v in sd.items()})
    with torch.no_grad():
        a = m.modality_encoders["AUDIO"].alibi_scale                   # [2 layers, 1, 12, 1, 1]; layer index 1 = blocks[0]
        a[1, 0, :, 0, 0] = scale
    md = m.to("cuda:0")
    Np = (N + 3) // 4 * 4
    for use_alibi, key in ((True, "attn_out_alibi"), (False, "attn_out_plain")):
        if not use_alibi:
            with torch.no_grad():
                md.modality_encoders["AUDIO"].alibi_scale.zero_()          # slope * 0: no positional bias
        for b in range(B):
            xp = torch.zeros((Np, dim), device="cuda:0")
            xp[:N] = x[b].to("cuda:0")
            with torch.no_grad():
                out = md._attention(xp, md.blocks[0].attn, 1, N, Np, 1)[:N].cpu()
            ref = torch.from_numpy(g[key][b])
            err = float((out - ref).abs().max() / ref.abs().max())
            assert err < 1e-4, (key, b, err)


@pytest.mark.parametrize("T,H,with_alibi", [(57, 12, True), (1535, 12, True), (64, 3, False), (33, 2, True), (200, 1, True)])
def test_flash_attention_matches_float64_softmax_attention(T, H, with_alibi):
    """avdf_flash_attn_alibi (all heads of a layer, scores on chip) against softmax(q k^T / sqrt(64) - slope |i - j|) v in
    float64, incl. a length that is not a multiple of the 32-key / 64-query tiles and padded rows that must stay untouched."""
    from audio_visual_deepfake_detection_amd._lib import lib, check
    hd, D = 64, 64 * H
    Tp = (T + 3) // 4 * 4
    g = torch.Generator().manual_seed(T * 7 + H)
    q, k, v = (torch.randn(Tp, D, generator=g) for _ in range(3))
    slopes = torch.rand(H, generator=g) * 0.5 if with_alibi else torch.zeros(H)
    qd, kd, vd = (t[:T].double().view(T, H, hd).permute(1, 0, 2) for t in (q, k, v))
    pos = -(torch.arange(T)[None, :] - torch.arange(T)[:, None]).abs().double()
    s = qd @ kd.transpose(1, 2) * hd ** -0.5 + slopes.double()[:, None, None] * pos[None]
    ref = (s.softmax(-1) @ vd).permute(1, 0, 2).reshape(T, D)
    out = torch.full((Tp, D), 7.0, device="cuda:0")
    qg, kg, vg, sg = q.cuda(), k.cuda(), v.cuda(), slopes.cuda()
    check(lib.avdf_flash_attn_alibi(qg.data_ptr(), kg.data_ptr(), vg.data_ptr(), out.data_ptr(), sg.data_ptr() if with_alibi else None,
                                    1, T, H, hd, D, Tp * D, hd ** -0.5, torch.cuda.current_stream().cuda_stream), "flash")
    torch.cuda.synchronize()
    err = float((out[:T].cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err
    assert float((out[T:] - 7.0).abs().max()) == 0.0 if Tp > T else True
