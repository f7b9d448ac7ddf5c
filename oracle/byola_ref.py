"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the BYOL-A extraction path.

PARITY UNPINNED: the reference's front-end is torchaudio.transforms.MelSpectrogram (torchaudio is not installed
here and no fixture exists) and its weights file is absent; this oracle restates the call site's parameters
(audio_feature/content_audio/extract_audio_feature_batch.py:42-60, config.yaml:4-10) on torch.stft and the network
(byol_a/models.py:53-85) on plain torch ops.  It pins the HIP path against torch, not against torchaudio.
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
