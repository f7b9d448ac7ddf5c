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


def test_emotion2vec_alibi_and_attention_oracle_match_reference(golden_dir):
    """oracle/emotion2vec_ref.{alibi_bias, alt_attention} == the reference's own get_alibi / get_alibi_bias / AltAttention
    (upstream/models/base.py:538-642, modules.py:252-322), executed by oracle/make_golden.py emotion2vec_parts."""
    import os
    import numpy as np
    g = np.load(os.path.join(golden_dir, "emotion2vec_parts.npz"))
    for heads, T in ((12, 33), (8, 17), (6, 9)):
        assert np.array_equal(R.alibi_bias(T, heads).numpy(), g[f"alibi_h{heads}_t{T}"])
    assert np.array_equal(R.alibi_bias(21, 12).numpy(), g["alibi_bias_b2_t21"][1])
    B, N, dim, heads = [int(v) for v in g["attn_shape"]]
    sd, x, scale = attn_fixture_inputs(g)
    bias = R.alibi_bias(N, heads) * scale.view(heads, 1, 1)
    out_b = R.alt_attention(sd, "", x, heads, bias)
    out_n = R.alt_attention(sd, "", x, heads, None)
    assert float((out_b - torch.from_numpy(g["attn_out_alibi"])).abs().max()) < 2e-6
    assert float((out_n - torch.from_numpy(g["attn_out_plain"])).abs().max()) < 2e-6


def attn_fixture_inputs(g):
    """Seeded AltAttention weights / input / per-head ALiBi scale of tests/golden/emotion2vec_parts.npz (same draws, in the
    same order, as oracle/make_golden.py case_emotion2vec_parts)."""
    import numpy as np
    B, N, dim, heads = [int(v) for v in g["attn_shape"]]
    rs = np.random.RandomState(int(g["attn_wseed"]))
    sd = {"attn.qkv.weight": rs.uniform(-1, 1, (3 * dim, dim)) / np.sqrt(dim) * 2.0, "attn.qkv.bias": 0.1 * rs.standard_normal(3 * dim),
          "attn.proj.weight": rs.uniform(-1, 1, (dim, dim)) / np.sqrt(dim), "attn.proj.bias": 0.1 * rs.standard_normal(dim)}
    sd = {k: torch.from_numpy(v.astype(np.float32)) for k, v in sd.items()}
    x = torch.from_numpy(np.random.RandomState(int(g["attn_xseed"])).standard_normal((B, N, dim)).astype(np.float32))
    scale = torch.from_numpy((0.5 + rs.uniform(0, 1, heads)).astype(np.float32))
    assert np.array_equal(scale.numpy(), g["attn_scale"])
    return sd, x, scale


def test_mel_filterbank_known_answers_of_the_htk_formula():
    """The log-mel front-end's filterbank (torchaudio melscale_fbanks, mel_scale='htk', norm=None -- torchaudio itself is
    absent) against answers worked out by hand from the published formula mel(f) = 2595 log10(1 + f/700): band edges,
    the first non-zero bin, the partition-of-unity property of un-normalised triangles, and peak positions.  Holds for
    the oracle's filterbank AND the product's (the constant the HIP GEMM consumes)."""
    import math
    import numpy as np
    from oracle import byola_ref as BR
    from audio_visual_deepfake_detection_amd.extractors.byola import melscale_fbanks_htk
    mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    inv = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    m_lo, m_hi = mel(60.0), mel(7800.0)
    # by hand: log10(1 + 60/700) = log10(1.085714) = 0.0357155 -> 92.682; log10(1 + 7800/700) = log10(12.142857) = 1.0843209 -> 2813.81
    assert abs(m_lo - 92.682) < 1e-3 and abs(m_hi - 2813.81) < 1e-2
    edges = [inv(m_lo + (m_hi - m_lo) * i / 65.0) for i in range(66)]                # 64 triangles share 66 edges
    assert abs(edges[0] - 60.0) < 1e-9 and abs(edges[65] - 7800.0) < 1e-8
    # second edge by hand: mel = 92.682 + (2813.813 - 92.682) / 65 = 134.546 -> 700 * (10^(134.546 / 2595) - 1) = 700 * 0.126803 = 88.76 Hz
    assert abs(edges[1] - 88.76) < 0.01
    freqs = [k * 8000.0 / 512.0 for k in range(513)]                                 # n_fft 1024 at 16 kHz: 15.625 Hz bins
    for name, fb in (("oracle", BR.mel_fbanks().numpy()), ("product", melscale_fbanks_htk(513, 60.0, 7800.0, 64, 16000))):
        assert fb.shape == (513, 64) and fb.dtype == np.float32
        assert float(np.abs(fb[:4]).max()) == 0.0, name                              # bins 0..46.9 Hz lie below f_min = 60 Hz
        want = (62.5 - 60.0) / (edges[1] - 60.0)                                     # bin 4 = 62.5 Hz on the first up-slope
        assert abs(float(fb[4, 0]) - want) < 1e-6 and float(np.abs(fb[4, 1:]).max()) == 0.0, name
        assert float(fb.max()) <= 1.0 + 1e-6 and float(fb.min()) >= 0.0
        inside = [k for k, f in enumerate(freqs) if edges[1] <= f <= edges[64]]
        assert np.allclose(fb[inside].sum(axis=1), 1.0, atol=2e-6), name             # down-slope(i) + up-slope(i+1) = 1
        assert float(np.abs(fb[[k for k, f in enumerate(freqs) if f >= 7800.0]]).max()) == 0.0, name
        for j in (0, 17, 40, 63):                                                    # triangle j: edges j, j+1 (peak), j+2
            k_peak = int(np.argmax(fb[:, j]))
            assert abs(freqs[k_peak] - edges[j + 1]) <= 15.625 / 2 + 1e-9, (name, j)
            k = next(i for i, f in enumerate(freqs) if edges[j] < f < edges[j + 1])
            assert abs(float(fb[k, j]) - (freqs[k] - edges[j]) / (edges[j + 1] - edges[j])) < 1e-6, (name, j)
