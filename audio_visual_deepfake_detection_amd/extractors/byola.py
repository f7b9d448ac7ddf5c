"""BYOL-A content-audio features on MI355X.

Replaces audio_feature/content_audio/extract_audio_feature_batch.py (log-mel front-end, :42-60) and
audio_feature/content_audio/byol_a/models.py:48-85 (AudioNTT2020Task6) of the reference.  Same module /
state-dict layout as the reference network (`features.{0,1,4,5,8,9}.*`, `fc.{0,3}.*`), so
`AudioNTT2020-BYOLA-64x96d2048.pth` loads unchanged (`load_weight` key filter of models.py:20-41 included).

Inference only (the extractor is frozen in the reference pipeline).  PARITY UNPINNED upstream: torchaudio and
the checkpoint are absent from the reference tree, so tests compare against a torch CPU restatement
(oracle/byola_ref.py) on random-init weights.
"""
import math
import re

import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import lib, check

SAMPLE_RATE, N_FFT, HOP, N_MELS, F_MIN, F_MAX = 16000, 1024, 160, 64, 60.0, 7800.0     # config.yaml:4-10
NORM_STATS = (-2.2800865, 3.5897882)                                                     # extract_audio_feature_batch.py:95


def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated (float64 maths -> fp32)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(np.float32)          # [n_freqs, n_mels]


class LogMelFrontEnd(nn.Module):
    """PCM [N] fp32 @16 kHz -> normalised log-mel [64, 1 + N // 160] on the device."""

    def __init__(self, stats=NORM_STATS):
        super().__init__()
        self.mean, self.std = float(stats[0]), float(stats[1])
        nb = N_FFT // 2 + 1                      # 513 bins
        self.kp = (nb + 3) // 4 * 4              # padded to 516 so the GEMM operands stay 16-byte aligned
        j = np.arange(N_FFT)[None, :]
        k = np.arange(nb)[:, None]
        ang = 2.0 * np.pi * k * j / N_FFT
        dft = np.zeros((2 * self.kp, N_FFT), dtype=np.float64)
        dft[:nb] = np.cos(ang)
        dft[self.kp:self.kp + nb] = -np.sin(ang)
        fb = np.zeros((N_MELS, self.kp), dtype=np.float32)
        fb[:, :nb] = melscale_fbanks_htk(nb, F_MIN, F_MAX, N_MELS, SAMPLE_RATE).T
        self.register_buffer("dft", torch.from_numpy(dft.astype(np.float32)).unsqueeze(-1).contiguous(), persistent=False)
        self.register_buffer("fb", torch.from_numpy(fb).unsqueeze(-1).contiguous(), persistent=False)
        self.register_buffer("window", torch.hann_window(N_FFT, periodic=True), persistent=False)

    @torch.no_grad()
    def forward(self, pcm):
        assert pcm.dim() == 1 and pcm.is_cuda and pcm.dtype == torch.float32
        n = pcm.numel()
        F_ = 1 + n // HOP
        st = torch.cuda.current_stream().cuda_stream
        frames = torch.empty((1, F_, N_FFT), dtype=torch.float32, device=pcm.device)
        check(lib.avdf_frame_window(pcm.contiguous().data_ptr(), self.window.data_ptr(), frames.data_ptr(), n, F_, N_FFT,
                                    HOP, st), "frame_window")
        spec = ops.conv1d(frames, self.dft)                                  # [1, F, 2*kp]  (re | im)
        power = torch.empty((1, F_, self.kp), dtype=torch.float32, device=pcm.device)
        check(lib.avdf_power_spectrum(spec.data_ptr(), power.data_ptr(), F_, self.kp, st), "power_spectrum")
        mel = ops.conv1d(power, self.fb)                                     # [1, F, 64]
        lms = torch.empty((N_MELS, F_), dtype=torch.float32, device=pcm.device)
        check(lib.avdf_logmel_norm(mel.data_ptr(), lms.data_ptr(), F_, N_MELS, float(torch.finfo(torch.float).eps),
                                   self.mean, self.std, st), "logmel_norm")
        return lms


class AudioNTT2020Task6(nn.Module):
    """DCASE2020 Task6 NTT audio embedding network, BYOL-A weights (byol_a/models.py:48-85)."""

    def __init__(self, n_mels=64, d=2048):
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(1, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2, stride=2))
        self.fc = nn.Sequential(nn.Linear(64 * (n_mels // (2 ** 3)), d), nn.ReLU(), nn.Dropout(p=0.3), nn.Linear(d, d),
                                nn.ReLU())
        self.d, self.n_mels = d, n_mels
        self._packed = None
        self.eval()

    def load_weight(self, weight_file, device, state_dict=None, key_check=True):
        """Same checkpoint key filter as the reference (byol_a/models.py:20-41)."""
        state_dict = state_dict or torch.load(weight_file, map_location=device)
        if 'state_dict' in state_dict:
            state_dict = state_dict['state_dict']
        if key_check:
            weights = {}
            for k in state_dict:
                m = re.search(r'(^fc\.|\.fc\.|^features\.|\.features\.)', k)
                if m is None:
                    continue
                nk = k[m.start():]
                weights[nk[1:] if nk[0] == '.' else nk] = state_dict[k]
        else:
            weights = state_dict
        self.load_state_dict(weights)
        self._packed = None
        return self.eval()

    def _pack(self):
        """Fold eval-mode BatchNorm into per-channel scale/shift (conv1) or into the GEMM weights (conv2/3) and
        re-index the 3x3 kernels for the mel-im2col GEMM: Wc[co, dm*64 + c, dt] = w[co, c, dm, dt]."""
        f = self.features
        pk = {}
        def bn_fold(conv, bn):
            s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            return s, bn.bias - bn.running_mean * s + conv.bias * s
        s, sh = bn_fold(f[0], f[1])
        pk["w1"] = f[0].weight.reshape(64, 9).contiguous()
        pk["s1"], pk["b1"] = s.contiguous(), sh.contiguous()
        for name, ci, bi in (("2", 4, 5), ("3", 8, 9)):
            s, sh = bn_fold(f[ci], f[bi])
            w = f[ci].weight * s[:, None, None, None]                     # [co, c, dm(mel), dt(time)]
            pk["w" + name] = w.permute(0, 2, 1, 3).reshape(64, 3 * 64, 3).contiguous()
            pk["b" + name] = sh.contiguous()
        self._packed = {k: v.detach() for k, v in pk.items()}

    @torch.no_grad()
    def forward_image(self, lms):
        """lms: [n_mels, Tm] normalised log-mel of ONE clip (device) -> [Tm // 8, d]."""
        assert not self.training, "the extractor is frozen (eval only)"
        if self._packed is None or self._packed["w1"].device != lms.device:
            self._pack()
        pk = self._packed
        M, T = lms.shape
        st = torch.cuda.current_stream().cuda_stream
        dev = lms.device
        a = torch.empty((M // 2, T // 2, 64), dtype=torch.float32, device=dev)
        check(lib.avdf_conv2d_c1_bn_relu_pool(lms.contiguous().data_ptr(), pk["w1"].data_ptr(), pk["s1"].data_ptr(),
                                              pk["b1"].data_ptr(), a.data_ptr(), M, T, 64, st), "conv2d_c1")
        for name, last in (("2", False), ("3", True)):
            Mc, Tc = a.shape[0], a.shape[1]
            col = torch.empty((Mc, Tc, 192), dtype=torch.float32, device=dev)
            check(lib.avdf_mel_im2col(a.data_ptr(), col.data_ptr(), Mc, Tc, 64, st), "mel_im2col")
            y = ops.conv1d(col, pk["w" + name], pk["b" + name], None, 1, ops.ACT_RELU)          # [Mc, Tc, 64]
            shp = (Tc // 2, Mc // 2, 64) if last else (Mc // 2, Tc // 2, 64)
            a = torch.empty(shp, dtype=torch.float32, device=dev)
            check(lib.avdf_maxpool2x2(y.data_ptr(), a.data_ptr(), Mc, Tc, 64, int(last), st), "maxpool2x2")
        Tq = a.shape[0]
        x = a.reshape(Tq, -1)                                # feature index = mel*64 + ch (models.py:81-83)
        x = ops.linear(x, self.fc[0].weight, self.fc[0].bias, ops.ACT_RELU)
        x = ops.linear(x, self.fc[3].weight, self.fc[3].bias, ops.ACT_RELU)
        return x

    @torch.no_grad()
    def forward(self, x):
        """Reference signature: x [B, 1, n_mels, Tm] -> [B, Tm // 8, d]."""
        return torch.stack([self.forward_image(x[b, 0]) for b in range(x.shape[0])])


@torch.no_grad()
def byola_features(pcm, model, frontend, pad_frames=None):
    """PCM [N] fp32 @16 kHz (device) -> [(1 + N // 160) // 8, 2048], the array the reference saves as .npy
    (extract_audio_feature_batch.py:104-123) before the dataset trims it (deepfake_video_audio.py:1683).
    pad_frames: log-mel length the clip is zero-padded to before the network (see `byola_features_batch`)."""
    lms = frontend(pcm)                      # STFT / mel GEMMs stay exact fp32: log() follows
    if pad_frames is not None and pad_frames > lms.shape[1]:
        lms = torch.nn.functional.pad(lms, (0, pad_frames - lms.shape[1]))          # zeros AFTER normalisation
    with ops.infer_mode():
        return model.forward_image(lms)


@torch.no_grad()
def byola_features_batch(pcms, model, frontend):
    """The reference's batch semantics (extract_audio_feature_batch.py:65-81,111-123): the NORMALISED log-mels of a batch
    are zero-padded to the longest one (post-normalisation zeros, i.e. not silence), the network runs on the padded
    images and every clip's feature array is saved UNTRIMMED ([max_frames // 8, 2048]; the dataset trims at load time,
    deepfake_video_audio.py:1683).  The network has no cross-clip operator (BatchNorm in eval mode), so clip i of the
    padded batch is the per-clip forward of its padded image; the rows of a shorter clip near and beyond its end differ
    from `byola_features(pcm)` because the 3x3 convs see relu(bn(0)) activations where the un-padded image ends."""
    frames = max(1 + int(p.numel()) // HOP for p in pcms)
    return [byola_features(p, model, frontend, pad_frames=frames) for p in pcms]
