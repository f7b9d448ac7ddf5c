"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BYOL-A extraction path.

NETWORK PINNED, FRONT-END UNPINNED: `byola_forward` (byol_a/models.py:53-85) is checked against the reference's own
AudioNTT2020Task6 class on seeded weights (oracle/make_golden.py byola_net -> tests/golden/byola_net.npz; that file
imports only torch).  The log-mel front-end is torchaudio.transforms.MelSpectrogram upstream (torchaudio is not
installed here, no fixture exists) and the weights file is absent: `log_mel` restates the call site's parameters
(audio_feature/content_audio/extract_audio_feature_batch.py:42-60, config.yaml:4-10) on torch.stft and pins the HIP
front-end against torch, not against torchaudio.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def mel_fbanks(n_freqs=513, f_min=60.0, f_max=7800.0, n_mels=64, sr=16000):
    all_freqs = torch.linspace(0, sr // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0).float()


def log_mel(pcm, stats=(-2.2800865, 3.5897882)):
    """pcm [N] -> [64, 1 + N//160]: MelSpectrogram defaults (power 2, centre, reflect, periodic Hann, HTK)."""
    spec = torch.stft(pcm, n_fft=1024, hop_length=160, win_length=1024, window=torch.hann_window(1024),
                      center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                       # [513, F]
    mel = mel_fbanks().t() @ power                               # [64, F]
    lms = (mel + torch.finfo(torch.float).eps).log()
    return (lms - stats[0]) / stats[1]


def byola_forward(sd, lms):
    """AudioNTT2020Task6.forward (byol_a/models.py:79-85), eval mode.  lms [B,1,64,T] -> [B, T//8, d]."""
    x = lms
    for c, b in ((0, 1), (4, 5), (8, 9)):
        x = F.conv2d(x, sd[f"features.{c}.weight"], sd[f"features.{c}.bias"], padding=1)
        x = F.batch_norm(x, sd[f"features.{b}.running_mean"], sd[f"features.{b}.running_var"],
                         sd[f"features.{b}.weight"], sd[f"features.{b}.bias"], training=False, eps=1e-5)
        x = F.max_pool2d(F.relu(x), 2, 2)
    x = x.permute(0, 3, 2, 1)
    B, T, D, C = x.shape
    x = x.reshape(B, T, C * D)
    x = F.relu(F.linear(x, sd["fc.0.weight"], sd["fc.0.bias"]))
    return F.relu(F.linear(x, sd["fc.3.weight"], sd["fc.3.bias"]))


BYOLA_KEYS = [("features.0", 64, 1), ("features.4", 64, 64), ("features.8", 64, 64)]


def seeded_state_dict(seed, n_mels=64, d=2048):
    """Deterministic AudioNTT2020Task6 state dict (reference key names) from numpy's legacy RandomState: fixtures carry
    the seed, not the weights.  BatchNorm statistics are non-trivial on purpose."""
    rs = np.random.RandomState(seed)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    sd = {}
    for (name, co, ci), bn in zip(BYOLA_KEYS, (1, 5, 9)):
        sd[f"{name}.weight"] = f32(rs.randn(co, ci, 3, 3) * (2.0 / (ci * 9)) ** 0.5)
        sd[f"{name}.bias"] = f32(rs.randn(co) * 0.1)
        sd[f"features.{bn}.weight"] = f32(1.0 + 0.1 * rs.randn(co))
        sd[f"features.{bn}.bias"] = f32(0.1 * rs.randn(co))
        sd[f"features.{bn}.running_mean"] = f32(0.3 * rs.randn(co))
        sd[f"features.{bn}.running_var"] = f32(rs.uniform(0.5, 1.5, co))
        sd[f"features.{bn}.num_batches_tracked"] = torch.tensor(1, dtype=torch.long)
    fin = 64 * (n_mels // 8)
    sd["fc.0.weight"] = f32(rs.randn(d, fin) * (2.0 / fin) ** 0.5)
    sd["fc.0.bias"] = f32(rs.randn(d) * 0.1)
    sd["fc.3.weight"] = f32(rs.randn(d, d) * (2.0 / d) ** 0.5)
    sd["fc.3.bias"] = f32(rs.randn(d) * 0.1)
    return sd
