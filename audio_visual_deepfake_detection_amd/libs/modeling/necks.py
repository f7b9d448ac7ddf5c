"""FPN neck on MI355X (reference: libs/modeling/necks.py:10-93)."""
from torch import nn

from .models import register_neck
from .blocks import MaskedConv1D, LayerNorm
from ... import ops


@register_neck("fpn")
class FPN1D(nn.Module):
    """lateral 1x1 (MFMA GEMM) -> top-down nearest x2 add -> depthwise k3 -> LayerNorm."""

    def __init__(self, in_channels, out_channel, scale_factor=2.0, start_level=0, end_level=-1, with_ln=True):
        super().__init__()
        assert isinstance(in_channels, (list, tuple))
        assert with_ln, "fpn_with_ln is True in every shipped config"
        self.in_channels, self.out_channel, self.scale_factor = in_channels, out_channel, scale_factor
        self.start_level = start_level
        self.end_level = len(in_channels) if end_level == -1 else end_level
        assert self.end_level <= len(in_channels) and 0 <= self.start_level < self.end_level
        self.lateral_convs, self.fpn_convs, self.fpn_norms = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(self.start_level, self.end_level):
            self.lateral_convs.append(MaskedConv1D(in_channels[i], out_channel, 1, bias=(not with_ln)))
            self.fpn_convs.append(MaskedConv1D(out_channel, out_channel, 3, padding=1, bias=(not with_ln),
                                               groups=out_channel))
            self.fpn_norms.append(LayerNorm(out_channel))

    def _laterals(self, inputs, fpn_masks):
        return [self.lateral_convs[i](inputs[i + self.start_level], fpn_masks[i + self.start_level])[0]
                for i in range(len(self.lateral_convs))]

    def _tail(self, laterals, fpn_masks):
        r = int(self.scale_factor)
        for i in range(len(laterals) - 1, 0, -1):
            laterals[i - 1] = ops.add_upsampled(laterals[i - 1], laterals[i], r)
        feats, masks = tuple(), tuple()
        for i in range(len(laterals)):
            x, m = self.fpn_convs[i](laterals[i], fpn_masks[i + self.start_level])
            feats += (self.fpn_norms[i](x),)
            masks += (m,)
        return feats, masks

    def forward_gapped(self, inputs, fpn_masks):
        """The levels as the shared heads consume them.  -> (x [B, sum(T_l + 1), C], mask [B, sum(T_l + 1)], lengths) when the
        fused tail applies (ops.fpn_tail_fusable: inference / engine path): the top-down adds, depthwise convs and LayerNorms
        of ALL levels are two launches instead of three per level and the concatenation disappears; otherwise
        (feats, masks) per level like forward()."""
        assert len(inputs) == len(self.in_channels) and len(fpn_masks) == len(self.in_channels)
        laterals = self._laterals(inputs, fpn_masks)
        dws = [c.conv.weight for c in self.fpn_convs]
        gammas, betas = [n.weight for n in self.fpn_norms], [n.bias for n in self.fpn_norms]
        if (int(self.scale_factor) == 2 and self.start_level == 0 and all(n.affine for n in self.fpn_norms)
                and len({n.eps for n in self.fpn_norms}) == 1 and ops.fpn_tail_fusable(laterals, dws, gammas, betas)):
            x, gmask = ops.fpn_tail(laterals, [fpn_masks[i] for i in range(len(laterals))], dws, gammas, betas,
                                    self.fpn_norms[0].eps)
            return x, gmask, tuple(int(t.shape[1]) for t in laterals)
        return self._tail(laterals, fpn_masks)

    def forward(self, inputs, fpn_masks):
        assert len(inputs) == len(self.in_channels) and len(fpn_masks) == len(self.in_channels)
        return self._tail(self._laterals(inputs, fpn_masks), fpn_masks)
