"""Building blocks of the UMMAFormer path on MI355X.

Same class names, constructor arguments and state-dict keys as the reference's
libs/modeling/blocks.py, but every forward runs hand-written gfx950 kernels (through ..ops) on
TIME-MAJOR activations:

    x    : fp32 [B, T, C]   (reference: [B, C, T])
    mask : fp32 [B, T]      (reference: bool [B, 1, T])

Parameters keep the reference layouts (nn.Conv1d / nn.Linear / [1,C,1] holders), so reference
checkpoints load unchanged and `make_optimizer`'s isinstance rules (train_utils.py:73-111) still
classify every parameter.  The nn.Conv1d / nn.Linear members are parameter holders only; their
aten forward is never called.
"""
import math

import numpy as np
import torch
from torch import nn

from ... import ops


import threading as _threading

_tls = _threading.local()          # per thread: an evaluation thread must not clear the cache under a training thread's forward


def _mask_cache():
    c = getattr(_tls, "masks", None)
    if c is None:
        c = _tls.masks = {}
    return c


def reset_mask_cache():
    """Called at the start of every network forward: masks of one forward are memoised, never across forwards (the
    engine's static mask buffer is refilled in place between steps)."""
    _mask_cache().clear()


def _mask_key(mask, stride):
    return (mask.data_ptr(), tuple(mask.shape), tuple(mask.stride()), stride, mask._version)


def prime_mask_pyramid(mask, n_levels, stride=2):
    """All pyramid masks of a forward in ONE launch (ops.mask_pyramid) instead of one strided copy per level; later
    `downsample_mask` calls of the interpolator, the backbone branches and the FPN hit the cache."""
    if not mask.is_cuda or mask.shape[1] % (stride ** n_levels) != 0:
        return
    cur = mask
    for nxt in ops.mask_pyramid(mask, n_levels, stride):
        _mask_cache()[_mask_key(cur, stride)] = (cur, nxt)
        cur = nxt


def downsample_mask(mask, stride):
    """Nearest-neighbour mask downsampling of MaskedConv1D (blocks.py:51-55) == mask[:, ::stride].  Every strided block
    asks for the same pyramid masks (attention, pooling skip, MLP): one strided copy per (mask, stride) per forward."""
    if stride == 1:
        return mask
    key = _mask_key(mask, stride)
    cache = _mask_cache()
    hit = cache.get(key)
    if hit is None:
        hit = (mask, mask[:, ::stride].contiguous())          # keeps `mask` alive: its address cannot be recycled
        cache[key] = hit
    return hit[1]


class MaskedConv1D(nn.Module):
    """Masked 1-D convolution (blocks.py:13-63).  groups must be 1 (dense, MFMA implicit GEMM) or
    in_channels (depthwise k3)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, padding_mode='zeros'):
        super().__init__()
        assert (kernel_size % 2 == 1) and (kernel_size // 2 == padding)
        assert dilation == 1 and padding_mode == 'zeros'
        assert groups == 1 or (groups == in_channels == out_channels and kernel_size == 3)
        self.stride = stride
        self.depthwise = groups > 1
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias,
                              padding_mode)
        if bias:
            torch.nn.init.constant_(self.conv.bias, 0.)

    def forward(self, x, mask, act=ops.ACT_NONE, exact=False):
        B, T, C = x.shape
        assert T % self.stride == 0            # blocks.py:46
        out_mask = downsample_mask(mask, self.stride)
        if self.depthwise:
            assert self.conv.bias is None
            y = ops.dwconv(x, self.conv.weight, out_mask, self.stride)
        else:
            y = ops.conv1d(x, self.conv.weight, self.conv.bias, out_mask, self.stride, act, exact=exact)
        return y, out_mask


class LayerNorm(nn.Module):
    """LayerNorm over channels (blocks.py:70-112); weight/bias are [1, C, 1] like the reference."""

    def __init__(self, num_channels, eps=1e-5, affine=True, device=None, dtype=None):
        super().__init__()
        kw = {'device': device, 'dtype': dtype}
        self.num_channels, self.eps, self.affine = num_channels, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones([1, num_channels, 1], **kw))
            self.bias = nn.Parameter(torch.zeros([1, num_channels, 1], **kw))
        else:
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)

    def forward(self, x, relu=False, pe=None, mask=None, rowscale=None):
        assert x.shape[-1] == self.num_channels
        return ops.layernorm(x, self.weight, self.bias, self.eps, relu, pe, mask, rowscale)


def get_sinusoid_encoding(n_position, d_hid):
    """Sinusoid table [1, C, T] (blocks.py:116-127), float64 maths cast to fp32."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    hid = np.arange(d_hid)[None, :]
    table = pos / np.power(10000, 2 * (hid // 2) / d_hid)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.FloatTensor(table).unsqueeze(0).transpose(1, 2)


class Scale(nn.Module):
    """Learnable scalar (blocks.py:1374-1392); applied inside the fused regression-head kernel."""

    def __init__(self, init_value=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(init_value, dtype=torch.float32), requires_grad=True)


class AffineDropPath(nn.Module):
    """Per-channel scale + per-sample stochastic depth (blocks.py:1424-1439).  The scale and the
    keep factor are consumed by the fused residual kernel; this module only owns the parameter
    and draws the per-sample keep factors."""

    def __init__(self, num_dim, drop_prob=0.0, init_scale_value=1e-4):
        super().__init__()
        self.scale = nn.Parameter(init_scale_value * torch.ones((1, num_dim, 1)), requires_grad=True)
        self.drop_prob = drop_prob

    def sample_keep(self, batch, device):
        """floor(keep_prob + U[0,1)) / keep_prob per sample (blocks.py:1403-1409); None when inactive.
        `plan_drop_path` may have pre-drawn the factors of every module of a network in one launch."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        pre = getattr(self, "_planned_keep", None)
        if pre is not None:
            self._planned_keep = None
            return pre
        keep_prob = 1 - self.drop_prob
        m = (keep_prob + torch.rand((batch,), dtype=torch.float32, device=device)).floor_()
        return m / keep_prob


def plan_drop_path(root, batch, device):
    """Draw the per-sample keep factors of ALL active AffineDropPath modules under `root` with one
    torch.rand (3 launches per step instead of 4 per module)."""
    mods = [m for m in root.modules() if isinstance(m, AffineDropPath) and m.training and m.drop_prob > 0.0]
    if not mods:
        return
    cache = getattr(root, "_drop_path_kp", None)      # device constant: built once (not capturable in a hipGraph)
    key = tuple(m.drop_prob for m in mods)
    if cache is None or cache[0] != key or cache[1].device != device:
        cache = (key, torch.tensor([1.0 - q for q in key], dtype=torch.float32, device=device)[:, None])
        root._drop_path_kp = cache
    kp = cache[1]
    keep = ops.droppath_keep(torch.rand((len(mods), batch), dtype=torch.float32, device=device), kp)
    for i, m in enumerate(mods):
        m._planned_keep = keep[i]


class _MHCABase(nn.Module):
    """Shared parameter set + dataflow of MaskedMHCA / MaskedMMHCA / LocalMaskedMHCA /
    LocalMaskedMMHCA (blocks.py:199-313, 317-437, 441-781, 883-1224): depthwise k3 conv (+mask)
    -> LayerNorm for q, k, v; 1x1 projections; attention core; 1x1 output projection (+mask).
    window_size <= 1 selects the global core."""

    def __init__(self, n_embd, n_head, window_size=-1, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0,
                 proj_pdrop=0.0, use_rel_pe=False, use_time_weight=False):
        super().__init__()
        assert n_embd % n_head == 0
        assert attn_pdrop == 0.0 and proj_pdrop == 0.0, "dropout inside attention is 0 in every shipped config"
        assert (n_qx_stride == 1) or (n_qx_stride % 2 == 0)
        assert (n_kv_stride == 1) or (n_kv_stride % 2 == 0)
        assert n_qx_stride in (1, 2) and n_kv_stride in (1, 2)
        self.n_embd, self.n_head = n_embd, n_head
        self.n_channels = n_embd // n_head
        self.scale = 1.0 / math.sqrt(self.n_channels)
        self.window_size = window_size
        self.window_overlap = window_size // 2
        self.n_qx_stride, self.n_kv_stride = n_qx_stride, n_kv_stride
        # NB the query conv uses the kv stride too (blocks.py:240,492,934) -- kept
        stride = n_kv_stride
        self.query_conv = MaskedConv1D(n_embd, n_embd, 3, stride=stride, padding=1, groups=n_embd, bias=False)
        self.query_norm = LayerNorm(n_embd)
        self.key_conv = MaskedConv1D(n_embd, n_embd, 3, stride=stride, padding=1, groups=n_embd, bias=False)
        self.key_norm = LayerNorm(n_embd)
        self.value_conv = MaskedConv1D(n_embd, n_embd, 3, stride=stride, padding=1, groups=n_embd, bias=False)
        self.value_norm = LayerNorm(n_embd)
        self.key = nn.Conv1d(n_embd, n_embd, 1)
        self.query = nn.Conv1d(n_embd, n_embd, 1)
        self.value = nn.Conv1d(n_embd, n_embd, 1)
        self.proj = nn.Conv1d(n_embd, n_embd, 1)
        # local attention only (blocks.py:968-975): per-head relative position bias before the softmax, per-head time
        # weighting after it; the parameter names end in `rel_pe` / `time_weighting` for make_optimizer's no-decay rule
        self.use_rel_pe = bool(use_rel_pe) and window_size > 1
        self.use_time_weight = bool(use_time_weight) and window_size > 1
        if self.use_rel_pe:
            self.rel_pe = nn.Parameter(torch.zeros(1, 1, n_head, window_size))
            nn.init.trunc_normal_(self.rel_pe, std=(2.0 / n_embd) ** 0.5)
        if self.use_time_weight:
            self.time_weighting = nn.Parameter(torch.ones(1, n_head, window_size))

    def _branch(self, conv, norm, proj, x, mask):
        y, out_mask = conv(x, mask)
        return ops.conv1d(norm(y), proj.weight, proj.bias), out_mask

    def attend_prenorm(self, xs, lns, masks, kv_shift=0):
        """Block entry: pre-LayerNorms `lns` (ln1 x3 | lnq, lnk, lnv) + attention on raw inputs `xs`.  When the
        three branches share length and mask (every call site of the shipped backbones) the LN -> depthwise conv
        -> LN chains of q, k, v run as ONE fused launch (ops.qkv_chain); otherwise the per-op path.
        kv_shift != 0: x_k / x_v are given at THEIR pyramid level and stand for their nearest resampling to x_q's length
        (> 0: up by 2^s, < 0: every 2^-s-th row -- backbones.py:487-490); the fused launch reads them through a row map,
        the per-op path resamples them first."""
        x_q, x_k, x_v = xs
        C = x_q.shape[-1]
        if kv_shift:
            Tq = x_q.shape[1]
            Tsrc = (Tq >> kv_shift) if kv_shift > 0 else (Tq << -kv_shift)
            fusable = (ops.FUSE_QKV_CHAIN and C <= 256 and x_k is x_v and x_k.shape == (x_q.shape[0], Tsrc, C)
                       and (kv_shift < 0 or Tq % (1 << kv_shift) == 0))
            if not fusable:
                r = 1 << abs(kv_shift)
                x_k = ops.upsample(x_k, r) if kv_shift > 0 else ops.downsample(x_k, r)
                x_v = x_k if xs[2] is xs[1] else (ops.upsample(x_v, r) if kv_shift > 0 else ops.downsample(x_v, r))
                xs, kv_shift = (x_q, x_k, x_v), 0
        same_mask = all(m is masks[0] or (m.data_ptr() == masks[0].data_ptr() and m.shape == masks[0].shape)
                        for m in masks[1:])
        if (ops.FUSE_QKV_CHAIN and C <= 256 and (kv_shift or (x_k.shape == x_q.shape and x_v.shape == x_q.shape)) and same_mask
                and all(ln.affine for ln in lns) and len({ln.eps for ln in lns}) == 1
                and self.query_norm.eps == lns[0].eps and x_q.shape[1] % self.n_kv_stride == 0):
            out_mask = downsample_mask(masks[0], self.n_kv_stride)
            norms = (self.query_norm, self.key_norm, self.value_norm)
            pt = torch.is_grad_enabled() and x_q.requires_grad
            ys = ops.qkv_chain(
                xs, [(ln.weight, ln.bias) for ln in lns],
                [self.query_conv.conv.weight, self.key_conv.conv.weight, self.value_conv.conv.weight],
                [(n.weight, n.bias) for n in norms], out_mask, self.n_kv_stride, lns[0].eps, passthrough=pt,
                xshift=(0, kv_shift, kv_shift) if kv_shift else None)
            q, k, v = ops.conv1d_group3(ys[:3], (self.query.weight, self.key.weight, self.value.weight),
                                        (self.query.bias, self.key.bias, self.value.bias))
            # ys[3] (training): x_q handed back by the chain op for the caller's residual connection
            return (*self._core(q, k, v, out_mask, out_mask), ys[3] if pt else x_q)
        if kv_shift:                      # fused launch not applicable after all (masks / LayerNorm variants): resample first
            r = 1 << abs(kv_shift)
            x_k = ops.upsample(x_k, r) if kv_shift > 0 else ops.downsample(x_k, r)
            x_v = x_k
        return (*self._attend(lns[0](x_q), masks[0], lns[1](x_k), masks[1], lns[2](x_v), masks[2]), x_q)

    def _core(self, q, k, v, qx_mask, kv_mask):
        if self.window_size > 1:
            out = ops.local_attention(q, k, v, kv_mask, self.n_head, self.window_size,
                                      self.rel_pe if self.use_rel_pe else None,
                                      self.time_weighting if self.use_time_weight else None)
        else:
            out = ops.global_attention(q, k, v, kv_mask, self.n_head)
        out = ops.conv1d(out, self.proj.weight, self.proj.bias, qx_mask)
        return out, qx_mask

    def _attend(self, x_q, mask_q, x_k, mask_k, x_v, mask_v):
        if ops.BRANCH_STREAMS and x_q.is_cuda:
            # the q / k / v chains (depthwise conv -> LayerNorm -> 1x1 projection) are independent and each is
            # far too small to fill 256 CUs: run them on three HIP streams (fork / join); autograd replays the
            # same streams in the backward pass and a captured hipGraph keeps the parallel branches
            cur = torch.cuda.current_stream()
            s1, s2 = ops.side_streams(x_q.device)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            q, qx_mask = self._branch(self.query_conv, self.query_norm, self.query, x_q, mask_q)
            with torch.cuda.stream(s1):
                k, kv_mask = self._branch(self.key_conv, self.key_norm, self.key, x_k, mask_k)
            with torch.cuda.stream(s2):
                v, _ = self._branch(self.value_conv, self.value_norm, self.value, x_v, mask_v)
            cur.wait_stream(s1)
            cur.wait_stream(s2)
            k.record_stream(cur)
            v.record_stream(cur)
            kv_mask.record_stream(cur)
        else:
            q, qx_mask = self._branch(self.query_conv, self.query_norm, self.query, x_q, mask_q)
            k, kv_mask = self._branch(self.key_conv, self.key_norm, self.key, x_k, mask_k)
            v, _ = self._branch(self.value_conv, self.value_norm, self.value, x_v, mask_v)
        return self._core(q, k, v, qx_mask, kv_mask)


class MaskedMHCA(_MHCABase):
    def __init__(self, n_embd, n_head, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__(n_embd, n_head, -1, n_qx_stride, n_kv_stride, attn_pdrop, proj_pdrop)

    def forward(self, x, mask):
        return self._attend(x, mask, x, mask, x, mask)


class LocalMaskedMHCA(_MHCABase):
    def __init__(self, n_embd, n_head, window_size, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0,
                 use_rel_pe=False, use_time_weight=False):
        assert window_size > 1 and n_head >= 1
        super().__init__(n_embd, n_head, window_size, n_qx_stride, n_kv_stride, attn_pdrop, proj_pdrop, use_rel_pe,
                         use_time_weight)

    def forward(self, x, mask):
        return self._attend(x, mask, x, mask, x, mask)


class MaskedMMHCA(_MHCABase):
    def __init__(self, n_embd, n_head, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0):
        super().__init__(n_embd, n_head, -1, n_qx_stride, n_kv_stride, attn_pdrop, proj_pdrop)

    def forward(self, x_q, mask_q, x_k=None, mask_k=None, x_v=None, mask_v=None):
        x_k, mask_k = (x_q, mask_q) if x_k is None else (x_k, mask_k)
        x_v, mask_v = (x_q, mask_q) if x_v is None else (x_v, mask_v)
        return self._attend(x_q, mask_q, x_k, mask_k, x_v, mask_v)


class LocalMaskedMMHCA(_MHCABase):
    def __init__(self, n_embd, n_head, window_size, n_qx_stride=1, n_kv_stride=1, attn_pdrop=0.0, proj_pdrop=0.0,
                 use_rel_pe=False, use_time_weight=False):
        assert window_size > 1 and n_head >= 1
        super().__init__(n_embd, n_head, window_size, n_qx_stride, n_kv_stride, attn_pdrop, proj_pdrop, use_rel_pe,
                         use_time_weight)

    def forward(self, x_q, mask_q, x_k=None, mask_k=None, x_v=None, mask_v=None):
        assert x_k is not None and x_v is not None, "every call site passes k and v (backbones.py:467,487,490)"
        return self._attend(x_q, mask_q, x_k, mask_k, x_v, mask_v)


class _BlockBase(nn.Module):
    def _build_common(self, n_embd, n_head, n_ds_strides, n_out, n_hidden, proj_pdrop, path_pdrop):
        assert len(n_ds_strides) == 2
        assert proj_pdrop == 0.0, "train_cfg['dropout'] is 0 in every shipped config"
        self.stride = n_ds_strides[0]
        assert self.stride in (1, 2), "pool_skip is MaxPool1d(3, 2, 1) for stride 2 (blocks.py:1277-1281)"
        self.pool_skip = nn.Identity()     # parameter-free; the max-pool runs in the HIP kernel
        n_hidden = 4 * n_embd if n_hidden is None else n_hidden
        n_out = n_embd if n_out is None else n_out
        self.mlp = nn.Sequential(nn.Conv1d(n_embd, n_hidden, 1), nn.GELU(), nn.Dropout(proj_pdrop, inplace=True),
                                 nn.Conv1d(n_hidden, n_out, 1), nn.Dropout(proj_pdrop, inplace=True))
        if path_pdrop > 0.0:
            self.drop_path_attn = AffineDropPath(n_embd, drop_prob=path_pdrop)
            self.drop_path_mlp = AffineDropPath(n_out, drop_prob=path_pdrop)
        else:
            self.drop_path_attn = nn.Identity()
            self.drop_path_mlp = nn.Identity()
            self.register_buffer("_ones", torch.ones(n_embd), persistent=False)

    def _residual_mlp(self, x, out, out_mask):
        """blocks.py:1311-1313 / 870-872: skip*mask + dp(attn); then + dp(mlp(ln2(.))*mask)."""
        B = x.shape[0]
        skip = ops.maxpool3s2(x) if self.stride > 1 else x
        if isinstance(self.drop_path_attn, AffineDropPath):
            sa, ka = self.drop_path_attn.scale, self.drop_path_attn.sample_keep(B, x.device)
            sm, km = self.drop_path_mlp.scale, self.drop_path_mlp.sample_keep(B, x.device)
        else:
            sa = sm = self._ones
            ka = km = None
        if self.ln2.affine:
            out, h = ops.residual_ln(skip, out_mask, out, None, sa, ka, self.ln2.weight, self.ln2.bias, self.ln2.eps)
        else:
            out = ops.affine_residual(skip, out_mask, out, None, sa, ka)
            h = self.ln2(out)
        if ops.mlp_residual_fusable(h, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias, out):
            return ops.mlp_residual(h, self.mlp[0].weight, self.mlp[0].bias, self.mlp[3].weight, self.mlp[3].bias, out_mask,
                                    out, sm, km)
        h = ops.conv1d(h, self.mlp[0].weight, self.mlp[0].bias, None, 1, ops.ACT_GELU)
        if ops.FUSE_FC2_RESIDUAL and self.mlp[3].weight.shape[0] == out.shape[-1]:
            return ops.conv1d_residual(h, self.mlp[3].weight, self.mlp[3].bias, out_mask, out, sm, km)
        h = ops.conv1d(h, self.mlp[3].weight, self.mlp[3].bias, out_mask)
        return ops.affine_residual(out, None, h, None, sm, km)


class TransformerBlock(_BlockBase):
    """Pre-LN block with (local) conv attention (blocks.py:1227-1317)."""

    def __init__(self, n_embd, n_head, n_ds_strides=(1, 1), n_out=None, n_hidden=None, act_layer=nn.GELU,
                 attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, mha_win_size=-1, use_rel_pe=False,
                 use_time_weight=False):
        super().__init__()
        assert act_layer is nn.GELU
        self.ln1 = LayerNorm(n_embd)
        self.ln2 = LayerNorm(n_embd)
        if mha_win_size > 1:
            self.attn = LocalMaskedMHCA(n_embd, n_head, window_size=mha_win_size, n_qx_stride=n_ds_strides[0],
                                        n_kv_stride=n_ds_strides[1], attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop,
                                        use_rel_pe=use_rel_pe, use_time_weight=use_time_weight)
        else:
            self.attn = MaskedMHCA(n_embd, n_head, n_qx_stride=n_ds_strides[0], n_kv_stride=n_ds_strides[1],
                                   attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop)
        self._build_common(n_embd, n_head, n_ds_strides, n_out, n_hidden, proj_pdrop, path_pdrop)

    def forward(self, x, mask, pos_embd=None):
        assert pos_embd is None
        out, out_mask, x_res = self.attn.attend_prenorm((x, x, x), (self.ln1, self.ln1, self.ln1), (mask, mask, mask))
        return self._residual_mlp(x_res, out, out_mask), out_mask


class MutilModelTransformerBlock(_BlockBase):
    """Cross-modal block: q from x_q, k from x_k, v from x_v (blocks.py:783-876)."""

    def __init__(self, n_embd, n_head, n_ds_strides=(1, 1), n_out=None, n_hidden=None, act_layer=nn.GELU,
                 attn_pdrop=0.0, proj_pdrop=0.0, path_pdrop=0.0, mha_win_size=-1, use_rel_pe=False,
                 use_time_weight=False):
        super().__init__()
        assert act_layer is nn.GELU
        self.lnq = LayerNorm(n_embd)
        self.lnk = LayerNorm(n_embd)
        self.lnv = LayerNorm(n_embd)
        self.ln2 = LayerNorm(n_embd)
        if mha_win_size > 1:
            self.attn = LocalMaskedMMHCA(n_embd, n_head, window_size=mha_win_size, n_qx_stride=n_ds_strides[0],
                                         n_kv_stride=n_ds_strides[1], attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop,
                                         use_rel_pe=use_rel_pe, use_time_weight=use_time_weight)
        else:
            self.attn = MaskedMMHCA(n_embd, n_head, n_qx_stride=n_ds_strides[0], n_kv_stride=n_ds_strides[1],
                                    attn_pdrop=attn_pdrop, proj_pdrop=proj_pdrop)
        self._build_common(n_embd, n_head, n_ds_strides, n_out, n_hidden, proj_pdrop, path_pdrop)

    def forward(self, x_q, mask_q, x_k=None, mask_k=None, x_v=None, mask_v=None, pos_embd=None, kv_shift=0):
        assert pos_embd is None
        out, out_mask, x_res = self.attn.attend_prenorm((x_q, x_k, x_v), (self.lnq, self.lnk, self.lnv),
                                                        (mask_q, mask_k, mask_v), kv_shift)
        return self._residual_mlp(x_res, out, out_mask), out_mask


class ConvBlock(nn.Module):
    """ResNet-style basic block (blocks.py:1320-1370): conv k3 (stride s) -> ReLU -> conv k3, + identity (1x1 stride-s conv
    when s > 1), ReLU.  The strided 1x1 conv is a row subsample followed by the 1x1 GEMM."""

    def __init__(self, n_embd, kernel_size=3, n_ds_stride=1, expansion_factor=2, n_out=None, act_layer=nn.ReLU):
        super().__init__()
        assert kernel_size == 3 and act_layer is nn.ReLU and n_ds_stride in (1, 2)
        n_out = n_embd if n_out is None else n_out
        width = n_embd * expansion_factor
        self.stride = n_ds_stride
        self.conv1 = MaskedConv1D(n_embd, width, kernel_size, n_ds_stride, padding=1)
        self.conv2 = MaskedConv1D(width, n_out, kernel_size, 1, padding=1)
        self.downsample = MaskedConv1D(n_embd, n_out, 1, 1) if n_ds_stride > 1 else None   # stride applied by subsampling
        self.register_buffer("_ones", torch.ones(n_out), persistent=False)

    def forward(self, x, mask, pos_embd=None):
        out, out_mask = self.conv1(x, mask, act=ops.ACT_RELU)
        out, out_mask = self.conv2(out, out_mask)
        identity = x
        if self.downsample is not None:
            identity, _ = self.downsample(ops.downsample(x, self.stride), out_mask)
        return torch.relu(ops.affine_residual(out, None, identity, None, self._ones)), out_mask


class DownBlock(nn.Module):
    """masked conv k3 (stride 2 by default) -> InstanceNorm1d -> LeakyReLU(0.2) (blocks.py:1495-1516)."""

    def __init__(self, in_channel, out_channel, kernel_size=3, stride=2, padding=1, dilation=1, norm=True):
        super().__init__()
        self.conv_block = MaskedConv1D(in_channel, out_channel, kernel_size, stride=stride, padding=padding,
                                       dilation=dilation)
        self.norm = norm
        assert norm, "every shipped config builds DownBlock with norm=True"

    def forward(self, x, mask):
        # feeds an InstanceNorm over time: training keeps the exact-fp32 GEMM (gradient parity);
        # the inference paths may use the split-bf16 one (the video-level logit stays within the 1e-3 bar on every golden)
        x, mask = self.conv_block(x, mask, exact=torch.is_grad_enabled())
        return ops.instnorm_lrelu(x, 1e-5, 0.2), mask


class MaskedConvTranspose1D(nn.Module):
    """Masked transposed conv (blocks.py:1443-1491): k3 / stride 2 / padding 1 / output_padding 1 doubles T;
    the mask is nearest-upsampled (each entry repeated).  Runs as the data-gradient of a stride-2 conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, dilation=1,
                 groups=1, padding_mode='zeros'):
        super().__init__()
        assert (kernel_size % 2 == 1) and (kernel_size // 2 == padding)
        self.stride = stride
        self.conv = nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride, padding,
                                       output_padding=output_padding, dilation=dilation, groups=groups,
                                       padding_mode=padding_mode)
        self.supported = (kernel_size == 3 and stride == 2 and padding == 1 and output_padding == 1 and groups == 1)

    def forward(self, x, mask):
        assert self.supported, "only the k3/s2/p1/op1 shape of Expansion (blocks.py:1571-1576) is implemented"
        out_mask = mask.repeat_interleave(self.stride, dim=1).contiguous()      # F.interpolate(nearest) to 2T
        return ops.conv_transpose1d_k3s2(x, self.conv.weight, self.conv.bias, out_mask), out_mask


class UpBlock(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size=3, padding=1, stride=2, dilation=1, output_padding=0,
                 norm=True, last=False):
        super().__init__()
        self.conv_transpose = MaskedConvTranspose1D(in_channel, out_channel, kernel_size, stride=stride,
                                                    padding=padding, dilation=dilation, output_padding=output_padding)
        self.norm, self.last = norm, last
        assert norm and not last, "Expansion(tanh=False) builds every UpBlock with IN + LeakyReLU (blocks.py:1599)"

    def forward(self, x, mask):
        x, mask = self.conv_transpose(x, mask)
        return ops.instnorm_lrelu(x, 1e-5, 0.2), mask


class Contraction(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_dims=256):
        super().__init__()
        self.down_1 = DownBlock(in_channels, hidden_dims)
        self.down_2 = DownBlock(hidden_dims, hidden_dims * 2)
        self.down_3 = DownBlock(hidden_dims * 2, hidden_dims * 4)
        self.down_4 = DownBlock(hidden_dims * 4, hidden_dims * 8)
        self.down_5 = DownBlock(hidden_dims * 8, out_channels)

    def forward(self, x, mask):
        for blk in (self.down_1, self.down_2, self.down_3, self.down_4, self.down_5):
            x, mask = blk(x, mask)
        return x, mask


class Expansion(nn.Module):
    def __init__(self, in_channels, out_channels, hidden_dims=2048, tanh=True):
        super().__init__()
        self.up_1 = UpBlock(in_channels, hidden_dims, output_padding=1)
        self.up_2 = UpBlock(hidden_dims, hidden_dims // 2, output_padding=1)
        self.up_3 = UpBlock(hidden_dims // 2, hidden_dims // 4, output_padding=1)
        self.up_4 = UpBlock(hidden_dims // 4, hidden_dims // 8, output_padding=1)
        self.up_5 = UpBlock(hidden_dims // 8, out_channels, output_padding=1, last=tanh)

    def forward(self, x, mask):
        for blk in (self.up_1, self.up_2, self.up_3, self.up_4, self.up_5):
            x, mask = blk(x, mask)
        return x, mask


class DeepInterpolator(nn.Module):
    """Contraction -> video-level classifier (blocks.py:1593-1638).  `norm=False` is the only mode the
    shipped meta-archs use (av_fd_no_recon.py:317).  The Expansion parameters exist for state-dict
    parity; its (unused) output is not computed."""

    def __init__(self, in_channels, hidden_channels=512, num_classes=1, norm=True):
        super().__init__()
        assert not norm, "normalize_batch is disabled by every shipped meta-arch"
        self.norm = norm
        self.contraction = Contraction(in_channels, hidden_channels)
        self.expansion = Expansion(hidden_channels, in_channels, tanh=False)
        self.conv0 = nn.Sequential(nn.Conv1d(hidden_channels, hidden_channels, kernel_size=1, bias=False),
                                   nn.InstanceNorm1d(hidden_channels), nn.LeakyReLU(0.2))
        self.conv1 = nn.Linear(hidden_channels * 2, hidden_channels, bias=False)
        self.conv2 = nn.Linear(hidden_channels, num_classes)
        self.bn1 = LayerNorm(hidden_channels)
        self.dp1 = nn.Dropout(p=0.5)

    def classifier(self, x):
        """blocks.py:1608-1618 on time-major [B, T, H] features -> [B, num_classes]."""
        B = x.shape[0]
        x = ops.conv1d(x, self.conv0[0].weight)
        x = ops.instnorm_lrelu(x, 1e-5, 0.2)
        pooled = ops.maxavgpool(x)                                   # [B, 2H] = cat(max, avg)
        h = ops.linear(pooled, self.conv1.weight)
        h = self.bn1(h, relu=True)
        if self.training and self.dp1.p > 0:
            keep = ops.dropout_keep(torch.rand_like(h), self.dp1.p)
            h = ops.scale_rows(h.reshape(-1, 1), keep.reshape(-1)).view_as(h)
        w = self.conv2.weight
        return ops.smallconv(h.view(B, 1, -1), ops.view_param(w, (w.shape[0], w.shape[1], 1)),
                             self.conv2.bias).view(B, -1)

    def forward(self, inputs, mask, need_reco=False):
        """-> video logits [B, num_classes] (and the reconstruction [B, T, C] when `need_reco`: only exp5
        consumes the Expansion output, av_fd_meta_arch.py:346-348; exp10/12 discard it)."""
        feat, fmask = self.contraction(inputs, mask)
        cls = self.classifier(feat)
        if not need_reco:
            return cls
        reco, _ = self.expansion(feat, fmask)
        return cls, reco


class Extract(nn.Module):
    """Five stride-1 DownBlocks C -> 1024 -> 512 -> 256 -> 128 -> 64 (blocks.py:1640-1661)."""

    def __init__(self, in_channels, hidden_dims=1024):
        super().__init__()
        self.down_1 = DownBlock(in_channels, hidden_dims, stride=1)
        self.down_2 = DownBlock(hidden_dims, hidden_dims // 2, stride=1)
        self.down_3 = DownBlock(hidden_dims // 2, hidden_dims // 4, stride=1)
        self.down_4 = DownBlock(hidden_dims // 4, hidden_dims // 8, stride=1)
        self.down_5 = DownBlock(hidden_dims // 8, hidden_dims // 16, stride=1)

    def forward(self, x, mask):
        for blk in (self.down_1, self.down_2, self.down_3, self.down_4, self.down_5):
            x, mask = blk(x, mask)
        return x, mask


class SegmentandCls(nn.Module):
    """Frame-level + video-level classifier of exp13 (blocks.py:1663-1721): Extract -> 1x1 conv + IN +
    LeakyReLU -> Linear(64->1) per frame; video logit = Linear(2->1) of (max, avg) over time."""

    def __init__(self, in_channels, hidden_channels=1024, num_classes=1, norm=True):
        super().__init__()
        assert not norm and num_classes == 1
        self.norm = norm
        self.contraction = Extract(in_channels, hidden_channels)
        c = hidden_channels // 16
        self.conv0 = nn.Sequential(nn.Conv1d(c, c, kernel_size=1, bias=False), nn.InstanceNorm1d(c), nn.LeakyReLU(0.2))
        self.cls_linear1 = nn.Linear(num_classes * 2, num_classes)
        self.seg_linear = nn.Linear(c, num_classes)
        self.bn1 = LayerNorm(hidden_channels)        # constructed by the reference, never used
        self.dp1 = nn.Dropout(p=0.5)

    def forward(self, inputs, mask):
        """-> (seg_scores [B, T], cls_scores [B, 1])"""
        B, T, _ = inputs.shape
        feat, _ = self.contraction(inputs, mask)
        x = ops.conv1d(feat, self.conv0[0].weight)
        x = ops.instnorm_lrelu(x, 1e-5, 0.2)
        w = self.seg_linear.weight
        seg = ops.smallconv(x, ops.view_param(w, (w.shape[0], w.shape[1], 1)), self.seg_linear.bias)     # [B,T,1]
        pooled = ops.maxavgpool(seg)                                                                   # [B,2] (max, avg)
        # Linear(2 -> 1): the narrow-conv kernel wants Cin % 4 == 0 -> zero-pad input and weight to 4
        pin = torch.nn.functional.pad(pooled, (0, 2)).view(B, 1, 4)
        pw = torch.nn.functional.pad(self.cls_linear1.weight, (0, 2)).view(1, 4, 1)
        cls = ops.smallconv(pin, pw, self.cls_linear1.bias).view(B, 1)
        return seg.view(B, T), cls
