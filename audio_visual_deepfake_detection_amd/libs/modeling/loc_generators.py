"""Temporal point generator (reference: libs/modeling/loc_generators.py:27-83)."""
import torch
from torch import nn

from .models import register_generator


class BufferList(nn.Module):
    def __init__(self, buffers):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b, persistent=False)     # not part of checkpoints

    def __len__(self):
        return len(self._buffers)

    def __iter__(self):
        return iter(self._buffers.values())


@register_generator('point')
class PointGenerator(nn.Module):
    """Per level [T_l, 4] = (t, reg_lo, reg_hi, stride)."""

    def __init__(self, max_seq_len, fpn_strides, regression_range, use_offset=False):
        super().__init__()
        assert len(regression_range) == len(fpn_strides)
        self.max_seq_len, self.fpn_levels = max_seq_len, len(fpn_strides)
        self.fpn_strides, self.regression_range, self.use_offset = fpn_strides, regression_range, use_offset
        pts = []
        for stride, (lo, hi) in zip(fpn_strides, regression_range):
            t = torch.arange(0, max_seq_len, stride, dtype=torch.float32)[:, None]
            if use_offset:
                t = t + 0.5 * stride
            pts.append(torch.cat((t, torch.full_like(t, float(lo)), torch.full_like(t, float(hi)),
                                  torch.full_like(t, float(stride))), dim=1))
        self.buffer_points = BufferList(pts)

    def forward(self, feat_lens):
        """feat_lens: per-level lengths (ints) or tensors whose time axis is dim 1 ([B, T_l, C])."""
        out = []
        for f, buf in zip(feat_lens, self.buffer_points):
            n = f if isinstance(f, int) else f.shape[1]
            assert n <= buf.shape[0], "Reached max buffer length for point generator"
            out.append(buf[:n, :])
        return out
