"""HR/LR multi-scale transformer backbone on MI355X (reference: libs/modeling/backbones.py:272-495)."""
import torch
from torch import nn

from .models import register_backbone
from .blocks import (MaskedConv1D, LayerNorm, TransformerBlock, MutilModelTransformerBlock, ConvBlock,
                     get_sinusoid_encoding, plan_drop_path)
from ... import ops


@register_backbone("convHRLRFullResSelfAttTransformerRevised")
class ConvHRLRFullResSelfAttTransformerBackboneRevised(nn.Module):
    """conv embedding -> cross-attn block -> stem -> 5 x (branch /2, lh_branch @ full res, hh_branch @ level).

    Differences from the reference forward that do not change any output:
      * the never-read `norm_x` embedding (backbones.py:441-442) is not computed;
      * when `reco_x is x` (exp10/exp12, av_fd_no_recon.py:349) its identical embedding is computed once;
      * hh_branch[last], whose output nobody consumes (backbones.py:488-490), is not evaluated
        (its parameters therefore get no gradient, exactly as in the reference).
    forward() takes time-major x [B,T,C] / fp32 mask [B,T] and returns 6 feats [B,T_l,D] + 6 masks [B,T_l].
    """

    def __init__(self, n_in, n_embd, n_head, n_embd_ks, max_len, arch=(2, 2, 5), mha_win_size=[-1] * 6,
                 scale_factor=2, with_ln=False, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, use_abs_pe=False,
                 use_rel_pe=False, use_time_weight=False):
        super().__init__()
        assert len(arch) == 3 and len(mha_win_size) == (1 + arch[2])
        assert not isinstance(n_in, (list, tuple)), "per-stream projection is unused by the shipped configs"
        assert with_ln, "embd_with_ln is True in every shipped config (libs/core/config.py:63)"
        self.n_in, self.arch, self.mha_win_size, self.max_len = n_in, arch, mha_win_size, max_len
        self.scale_factor, self.use_abs_pe = scale_factor, use_abs_pe
        self.proj = None
        self.embd = nn.ModuleList()
        self.embd_norm = nn.ModuleList()
        for idx in range(arch[0]):
            self.embd.append(MaskedConv1D(n_in if idx == 0 else n_embd, n_embd, n_embd_ks, stride=1,
                                          padding=n_embd_ks // 2, bias=(not with_ln)))
            self.embd_norm.append(LayerNorm(n_embd))
        if use_abs_pe:
            pos_embd = get_sinusoid_encoding(max_len, n_embd) / (n_embd ** 0.5)       # backbones.py:337
            self.register_buffer("pos_embd", pos_embd, persistent=False)
            self.register_buffer("pos_embd_tm", pos_embd[0].t().contiguous(), persistent=False)   # [T, C]
        blk = dict(attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop, path_pdrop=path_pdrop, use_rel_pe=use_rel_pe,
                   use_time_weight=use_time_weight)
        self.resselfattention = MutilModelTransformerBlock(n_embd, n_head, n_ds_strides=(1, 1),
                                                           mha_win_size=mha_win_size[0], **blk)
        self.stem = nn.ModuleList([TransformerBlock(n_embd, n_head, n_ds_strides=(1, 1), mha_win_size=mha_win_size[0],
                                                    **blk) for _ in range(arch[1])])
        self.branch, self.lh_branch, self.hh_branch = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for idx in range(arch[2]):
            self.branch.append(TransformerBlock(n_embd, n_head, n_ds_strides=(scale_factor, scale_factor),
                                                mha_win_size=mha_win_size[1 + idx], **blk))
            self.lh_branch.append(MutilModelTransformerBlock(n_embd, n_head, n_ds_strides=(1, 1),
                                                             mha_win_size=mha_win_size[0], **blk))
            self.hh_branch.append(MutilModelTransformerBlock(n_embd, n_head, n_ds_strides=(1, 1),
                                                             mha_win_size=mha_win_size[0], **blk))
        self.apply(self.__init_weights__)

    def __init_weights__(self, module):
        if isinstance(module, (nn.Linear, nn.Conv1d)) and module.bias is not None:
            torch.nn.init.constant_(module.bias, 0.)

    def _embed(self, x, mask):
        T = x.shape[1]
        last = len(self.embd) - 1
        for idx in range(len(self.embd)):
            x, mask = self.embd[idx](x, mask)
            pe = None
            if idx == last and self.use_abs_pe:
                # training: fixed-length table; eval: table linearly re-interpolated to T rows when T >= max_len
                # (backbones.py:447-465; identity at T == max_len) -- the K1 kernel is that interpolation
                if self.training or T <= self.max_len:
                    assert T <= self.max_len, "Reached max length."
                    pe = self.pos_embd_tm[:T]
                else:
                    pe = ops.interp_concat([self.pos_embd_tm], T)
            x = self.embd_norm[idx](x, relu=True, pe=pe, mask=mask if pe is not None else None)
        return x

    def forward(self, x, reco_x, mask, cut_hook=None):
        """`cut_hook` (engine.TrainStep): called on the stem output, the one tensor every later block depends on; the
        engine detaches there so that the backward pass runs in phases whose gradient buckets can be all-reduced while
        the next phase computes."""
        if self.training:
            plan_drop_path(self, x.shape[0], x.device)
        same = reco_x is None or reco_x is x
        x = self._embed(x, mask)
        reco_x = x if same else self._embed(reco_x, mask)
        x, _ = self.resselfattention(x, mask, x_k=reco_x, mask_k=mask, x_v=x, mask_v=mask)
        for blk in self.stem:
            x, mask = blk(x, mask)
        if cut_hook is not None:
            x = cut_hook(x)
        lh_feat, lh_mask = x, mask
        out_feats, out_masks = [lh_feat], [lh_mask]
        n = len(self.branch)
        for idx in range(n):
            x, mask = self.branch[idx](x, mask)
            r = self.scale_factor ** (idx + 1)
            if self.scale_factor == 2:
                # the nearest up / down-sampling of the reference (backbones.py:487,490) is a row map inside the k / v chain
                # kernels of the cross-level blocks: the resampled tensor is never materialised
                lh_feat, lh_mask = self.lh_branch[idx](lh_feat, lh_mask, x_k=x, mask_k=lh_mask, x_v=x, mask_v=lh_mask,
                                                       kv_shift=idx + 1)
            else:
                up = ops.upsample(x, r)                                    # F.interpolate nearest (backbones.py:487)
                lh_feat, lh_mask = self.lh_branch[idx](lh_feat, lh_mask, x_k=up, mask_k=lh_mask, x_v=up, mask_v=lh_mask)
            out_feats.append(x)
            out_masks.append(mask)
            if idx < n - 1:
                if self.scale_factor == 2:
                    x, mask = self.hh_branch[idx](x, mask, x_k=lh_feat, mask_k=mask, x_v=lh_feat, mask_v=mask,
                                                  kv_shift=-(idx + 1))
                else:
                    dn = ops.downsample(lh_feat, r)                        # backbones.py:490
                    x, mask = self.hh_branch[idx](x, mask, x_k=dn, mask_k=mask, x_v=dn, mask_v=mask)
        out_feats[0], out_masks[0] = lh_feat, lh_mask
        return out_feats, out_masks


@register_backbone("convTransformer")
class ConvTransformerBackbone(nn.Module):
    """ActionFormer backbone (reference: libs/modeling/backbones.py:10-168): conv embedding (+LN, ReLU, abs PE) ->
    stem TransformerBlocks -> branch TransformerBlocks with stride-`scale_factor` pooling; every branch output is a
    pyramid level.  forward(x [B,T,C] time-major, mask [B,T]) -> (feats, masks) tuples of arch[2]+1 levels."""

    def __init__(self, n_in, n_embd, n_head, n_embd_ks, max_len, arch=(2, 2, 5), mha_win_size=[-1] * 6,
                 scale_factor=2, with_ln=False, attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, use_abs_pe=False,
                 use_rel_pe=False, use_time_weight=False):
        super().__init__()
        assert len(arch) == 3 and len(mha_win_size) == (1 + arch[2])
        assert not isinstance(n_in, (list, tuple)), "per-stream projection is unused by the shipped configs"
        assert with_ln, "embd_with_ln is True in every shipped config (libs/core/config.py:63)"
        self.n_in, self.arch, self.mha_win_size, self.max_len = n_in, arch, mha_win_size, max_len
        self.scale_factor, self.use_abs_pe = scale_factor, use_abs_pe
        self.proj = None
        self.embd = nn.ModuleList()
        self.embd_norm = nn.ModuleList()
        for idx in range(arch[0]):
            self.embd.append(MaskedConv1D(n_in if idx == 0 else n_embd, n_embd, n_embd_ks, stride=1,
                                          padding=n_embd_ks // 2, bias=(not with_ln)))
            self.embd_norm.append(LayerNorm(n_embd))
        if use_abs_pe:
            pos_embd = get_sinusoid_encoding(max_len, n_embd) / (n_embd ** 0.5)
            self.register_buffer("pos_embd", pos_embd, persistent=False)
            self.register_buffer("pos_embd_tm", pos_embd[0].t().contiguous(), persistent=False)
        blk = dict(attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop, path_pdrop=path_pdrop, use_rel_pe=use_rel_pe,
                   use_time_weight=use_time_weight)
        self.stem = nn.ModuleList([TransformerBlock(n_embd, n_head, n_ds_strides=(1, 1), mha_win_size=mha_win_size[0],
                                                    **blk) for _ in range(arch[1])])
        self.branch = nn.ModuleList([TransformerBlock(n_embd, n_head, n_ds_strides=(scale_factor, scale_factor),
                                                      mha_win_size=mha_win_size[1 + idx], **blk)
                                     for idx in range(arch[2])])
        self.apply(self.__init_weights__)

    def __init_weights__(self, module):
        if isinstance(module, (nn.Linear, nn.Conv1d)) and module.bias is not None:
            torch.nn.init.constant_(module.bias, 0.)

    _embed = ConvHRLRFullResSelfAttTransformerBackboneRevised._embed

    def forward(self, x, mask, cut_hook=None):
        if self.training:
            plan_drop_path(self, x.shape[0], x.device)
        x = self._embed(x, mask)
        for blk in self.stem:
            x, mask = blk(x, mask)
        if cut_hook is not None:
            x = cut_hook(x)
        out_feats, out_masks = [x], [mask]
        for blk in self.branch:
            x, mask = blk(x, mask)
            out_feats.append(x)
            out_masks.append(mask)
        return out_feats, out_masks


@register_backbone("conv")
class ConvBackbone(nn.Module):
    """Convolution-only backbone (reference: libs/modeling/backbones.py:171-269): conv embedding (+LN, ReLU) -> stem
    ConvBlocks -> strided branch ConvBlocks, one pyramid level per branch block."""

    def __init__(self, n_in, n_embd, n_embd_ks, arch=(2, 2, 5), scale_factor=2, with_ln=False):
        super().__init__()
        assert len(arch) == 3 and not isinstance(n_in, (list, tuple)) and with_ln
        self.n_in, self.arch, self.scale_factor = n_in, arch, scale_factor
        self.use_abs_pe, self.proj = False, None
        self.embd = nn.ModuleList()
        self.embd_norm = nn.ModuleList()
        for idx in range(arch[0]):
            self.embd.append(MaskedConv1D(n_in if idx == 0 else n_embd, n_embd, n_embd_ks, stride=1,
                                          padding=n_embd_ks // 2, bias=(not with_ln)))
            self.embd_norm.append(LayerNorm(n_embd))
        self.stem = nn.ModuleList([ConvBlock(n_embd, 3, 1) for _ in range(arch[1])])
        self.branch = nn.ModuleList([ConvBlock(n_embd, 3, scale_factor) for _ in range(arch[2])])
        self.apply(self.__init_weights__)

    def __init_weights__(self, module):
        if isinstance(module, (nn.Linear, nn.Conv1d)) and module.bias is not None:
            torch.nn.init.constant_(module.bias, 0.)

    _embed = ConvHRLRFullResSelfAttTransformerBackboneRevised._embed

    def forward(self, x, mask, cut_hook=None):
        x = self._embed(x, mask)
        for blk in self.stem:
            x, mask = blk(x, mask)
        if cut_hook is not None:
            x = cut_hook(x)
        out_feats, out_masks = [x], [mask]
        for blk in self.branch:
            x, mask = blk(x, mask)
            out_feats.append(x)
            out_masks.append(mask)
        return out_feats, out_masks
