"""CPU checks of the frozen-extractor hosts: checkpoint key layout + output-length rules (no compute on HIP)."""
import torch

from audio_visual_deepfake_detection_amd.extractors import Emotion2Vec
from oracle import emotion2vec_ref as R


def test_emotion2vec_state_dict_uses_fairseq_keys():
    m = Emotion2Vec()
    keys = set(m.state_dict().keys())
    a = "modality_encoders.AUDIO."
    for k in [a + "local_encoder.conv_layers.0.0.weight", a + "local_encoder.conv_layers.6.2.1.bias",
              a + "project_features.1.weight", a + "project_features.2.bias",
              a + "relative_positional_encoder.1.0.weight", a + "relative_positional_encoder.5.0.bias",
              a + "context_encoder.blocks.3.attn.qkv.weight", a + "context_encoder.norm.weight", a + "alibi_scale",
              "blocks.7.mlp.fc2.bias", "blocks.0.norm1.weight"]:
        assert k in keys, k
    sd = m.state_dict()
    assert sd[a + "local_encoder.conv_layers.0.0.weight"].shape == (512, 1, 10)
    assert sd[a + "local_encoder.conv_layers.5.0.weight"].shape == (512, 512, 2)
    assert sd[a + "relative_positional_encoder.1.0.weight"].shape == (768, 48, 19)
    assert sd[a + "alibi_scale"].shape == (12, 1, 12, 1, 1)
    import re
    assert not any(re.search(r"relative_positional_encoder\.\d\.3\.", k) for k in keys)      # LN without affine


def test_emotion2vec_restatement_length_rule_and_alibi_slopes():
    m = Emotion2Vec(depth=1, prenet_depth=1)
    y = R.forward(dict(m.state_dict()), torch.randn(1, 4000) * 0.1, prenet_depth=1, depth=1)
    assert y.shape == (1, (4000 - 400) // 320 + 1, 768)
    s = R.slopes(12)
    assert len(s) == 12 and abs(s[0] - 2 ** -1) < 1e-12 and abs(s[7] - 2 ** -8) < 1e-12 and abs(s[8] - 2 ** -0.5) < 1e-12


def test_byola_oracle_matches_reference_network(golden_dir):
    """oracle/byola_ref.byola_forward == the reference's AudioNTT2020Task6 (byol_a/models.py:48-85) on the seeded weights
    and log-mel batch of tests/golden/byola_net.npz (written by oracle/make_golden.py byola_net from the real class)."""
    import os
    import numpy as np
    from oracle import byola_ref as BR
    from audio_visual_deepfake_detection_amd.extractors import AudioNTT2020Task6
    g = np.load(os.path.join(golden_dir, "byola_net.npz"))
    sd = BR.seeded_state_dict(int(g["wseed"]))
    lms = torch.from_numpy(np.random.RandomState(int(g["xseed"])).randn(*g["lms_shape"]).astype(np.float32))
    out = BR.byola_forward(sd, lms)
    ref = torch.from_numpy(g["out"])
    assert out.shape == ref.shape
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-6
    # the host module takes the reference checkpoint's keys as they are
    m = AudioNTT2020Task6(64, 2048)
    assert set(m.state_dict().keys()) == set(sd.keys())
    m.load_state_dict(sd, strict=True)
