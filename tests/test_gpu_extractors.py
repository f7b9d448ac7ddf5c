"""BYOL-A extraction path on a real MI355X vs the torch CPU restatement (oracle/byola_ref.py).
PARITY UNPINNED upstream (no torchaudio / checkpoint in the reference tree): this pins HIP == torch."""
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
