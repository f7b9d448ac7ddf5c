"""LAV-DF visual encoder (C3D) frame features on MI355X.

Replaces video_feature/aligned_video/model/video_encoder.py:12-62 (`C3DVideoEncoder`), `AlignVideo.forward_features`
(video_model.py:136-140) and the chunk / pad / resize / stitch logic of extract_video_features.py:19-77,136-166.

ASSUMPTION (parity unpinned): `utils.Conv3d/Conv1d` (video_encoder.py:9) are not in the reference tree; this module
follows the upstream LAV-DF definition `conv -> activation` (sub-module `conv`), with an optional BatchNorm (sub-module
`bn`, eval statistics folded into the conv at pack time) for checkpoints that carry one.  Parameter names therefore are
`block{0..3}.{i}.conv.{weight,bias}` as a `video_encoder.`-stripped AlignVideo checkpoint would hold them.
The MViT encoders (video_encoder.py:65-170) need torchvision internals that are absent: unsupported, raises.

Exact tail trimming: a chunk's padding frames are zeros, and each of the 8 convs sees +-1 frame, so output frames
[0, L) of the last chunk depend only on input frames [0, L+8): the last chunk is computed on min(512, L+8) frames
(bit-identical rows, ~25% less work on a 768-frame clip).  `trim_tail=False` restores the full 512-frame chunk.
"""
import os

import torch
from torch import nn

from .. import ops
from .._lib import lib, check

CHUNK = 512            # extract_video_features.py:42 frame_padding
SIDE = 96              # :58,67 resize target
_RF = 8                # temporal receptive-field radius of the conv stack (8 layers x 1 frame)


class _ConvNd(nn.Module):
    def __init__(self, conv, bn=None):
        super().__init__()
        self.conv = conv
        if bn is not None:
            self.bn = bn


class C3DVideoEncoder(nn.Module):
    def __init__(self, n_features=(64, 96, 128, 128), v_cla_feature_in=256, batch_norm=False):
        super().__init__()
        d0, d1, d2, d3 = n_features
        assert all(d % 4 == 0 for d in n_features)

        def c3(i, o):
            return _ConvNd(nn.Conv3d(i, o, 3, 1, 1), nn.BatchNorm3d(o) if batch_norm else None)

        I = nn.Identity
        self.block0 = nn.Sequential(c3(3, d0), c3(d0, d0), I())
        self.block1 = nn.Sequential(c3(d0, d1), c3(d1, d1), I())
        self.block2 = nn.Sequential(c3(d1, d2), c3(d2, d2), I())
        self.block3 = nn.Sequential(c3(d2, d3), I(), c3(d3, d3), I(), I(),
                                    _ConvNd(nn.Conv1d(d3 * 4, v_cla_feature_in, 1),
                                            nn.BatchNorm1d(v_cla_feature_in) if batch_norm else None))
        # (module, pool after it)
        self._plan = [(self.block0[0], 0), (self.block0[1], 3), (self.block1[0], 0), (self.block1[1], 2),
                      (self.block2[0], 0), (self.block2[1], 2), (self.block3[0], 2), (self.block3[2], 2)]
        self._packed = None
        self.eval()

    @staticmethod
    def _fold(m):
        w, b = m.conv.weight.detach().float(), m.conv.bias.detach().float()
        if hasattr(m, "bn"):
            s = m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)
            w = w * s.view(-1, *([1] * (w.dim() - 1)))
            b = (b - m.bn.running_mean) * s + m.bn.bias
        return w, b

    def pack(self):
        """One-time weight preparation: fold BN, tap-major [27, Cout, Cin(pad 4)] conv weights, and the
        `b c t h w -> b (c h w) t` rearrangement folded into the 1x1 conv's input columns."""
        packed = []
        for m, _ in self._plan:
            w, b = self._fold(m)                                   # [Cout, Cin, 3, 3, 3]
            cin = w.shape[1]
            if cin % 4:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, 4 - cin % 4))
            packed.append((w.permute(2, 3, 4, 0, 1).contiguous(), b.contiguous()))
        w, b = self._fold(self.block3[5])                          # [256, C*4, 1], columns ordered (c, h, w)
        o, k, _ = w.shape
        w = w.view(o, k // 4, 4).permute(0, 2, 1).reshape(o, k, 1).contiguous()      # -> (h, w, c)
        self._packed = (packed, (w, b.contiguous()))
        return self

    @torch.no_grad()
    def forward_prepared(self, x):
        """x [B, T, 96, 96, 4] fp32 voxel-major (avdf_video_prep output) -> [B, T, 256]."""
        if self._packed is None or self._packed[1][0].device != x.device:
            self.pack()
        st = torch.cuda.current_stream().cuda_stream
        B, T, H, W, _ = x.shape
        assert H == W and H % 48 == 0, "the pooling chain (3,2,2,2,2) must end at 2x2"
        bf3 = 1 if (ops.INFER_BF16X3 or (ops.current_gemm_mode() & 1)) else 0      # opt-in (exact fp32 by default)
        check(lib.avdf_conv3d_mode(bf3), "conv3d_mode")          # split-bf16 operands for the Cin >= 16 layers
        for (wp, b), (_, pool) in zip(self._packed[0], self._plan):
            cout, cin = wp.shape[3], wp.shape[4]
            y = torch.empty((B, T, H, W, cout), dtype=torch.float32, device=x.device)
            # cin of the packed weights is padded to 4 for the first layer: algorithmic FLOPs use the real 3 input channels
            fl = 2.0 * B * T * H * W * cout * (3 if cin == 4 else cin) * 27
            ops._timed("conv3d_k3", fl, lambda: check(
                lib.avdf_conv3d_k3(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), B, T, H, W, cin, cout,
                                   ops.ACT_LEAKY, st), "conv3d_k3"))
            x = y
            if pool:
                y = torch.empty((B, T, H // pool, W // pool, cout), dtype=torch.float32, device=x.device)
                check(lib.avdf_maxpool_hw(x.data_ptr(), y.data_ptr(), B * T, H, W, cout, pool, st), "maxpool_hw")
                x, H, W = y, H // pool, W // pool
        check(lib.avdf_conv3d_mode(0), "conv3d_mode")
        assert H * W == 4, (H, W)
        w1, b1 = self._packed[1]
        with ops.infer_mode():
            return ops.linear(x.view(B * T, -1), w1, b1, ops.ACT_LEAKY).view(B, T, -1)

    @torch.no_grad()
    def forward(self, video):
        """Reference signature: video [B, 3, T, H, W] fp32 in [0,1] -> [B, 256, T] (video_encoder.py:57-62)."""
        B, C, T, H, W = video.shape
        x = torch.zeros((B, T, H, W, 4), dtype=torch.float32, device=video.device)
        x[..., :3] = video.permute(0, 2, 3, 4, 1)
        return self.forward_prepared(x).permute(0, 2, 1)


def get_video_encoder(v_cla_feature_in=256, temporal_size=512, v_encoder="c3d", ve_features=(64, 96, 128, 128)):
    """video_encoder.py:184-201 selector, C3D only."""
    if v_encoder == "c3d":
        return C3DVideoEncoder(n_features=ve_features, v_cla_feature_in=v_cla_feature_in)
    raise ValueError(f"video encoder {v_encoder!r} is not supported on this path (needs torchvision MViT internals)")


def chunk_plan(n_frames, chunk=None):
    """extract_video_features.py:53-74,149-160: floor(len/512)+1 chunks; rows kept of the last one = len % 512
    (so a video of exactly k*512 frames gets an all-padding chunk contributing 0 rows)."""
    chunk = CHUNK if chunk is None else chunk
    n = n_frames // chunk + 1
    return [(i * chunk, min(chunk, max(0, n_frames - i * chunk)), chunk if i < n - 1 else n_frames % chunk)
            for i in range(n)]


@torch.no_grad()
def video_features(frames_u8, model, trim_tail=True):
    """uint8 frames [T, H0, W0, 3] (decoded video, device tensor) -> [T, 256] as the reference saves it."""
    assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 4 and frames_u8.shape[3] == 3 and frames_u8.is_cuda
    frames_u8 = frames_u8.contiguous()
    T, H0, W0, _ = frames_u8.shape
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for start, n_real, keep in chunk_plan(T):
        if keep == 0:
            continue                                         # all-padding chunk: no rows are kept from it
        tp = min(CHUNK, keep + _RF) if trim_tail else CHUNK
        x = torch.empty((1, tp, SIDE, SIDE, 4), dtype=torch.float32, device=frames_u8.device)
        check(lib.avdf_video_prep(frames_u8.data_ptr() + start * H0 * W0 * 3, x.data_ptr(), min(n_real, tp), H0, W0, tp,
                                  SIDE, st), "video_prep")
        out.append(model.forward_prepared(x)[0, :keep])
    return torch.cat(out, 0) if out else torch.zeros((0, 256), device=frames_u8.device)
