"""Emotion2Vec (data2vec 2.0 audio) frame features on MI355X.

Replaces audio_feature/emotion_audio/scripts/extract_audio_feature_batch.py:73-98 and the inference path of
audio_feature/emotion_audio/upstream/models (emotion2vec.py:97-188 `extract_features(mask=False)`, audio.py:52-138,
base.py:189-329,538-642, modules.py:63-97,184-322) plus fairseq's ConvFeatureExtractionModel (mode='layer_norm',
conv_bias=False; named at audio.py:12-17).  Module / parameter names mirror the fairseq checkpoint
(`modality_encoders.AUDIO.*`, `blocks.*`) so `emotion2vec_base.pt['model']` can be loaded with strict=False (decoder /
EMA entries are pre-training only).  Every config field that changes the inference graph is a constructor argument
because the real values live in the absent checkpoint's cfg.

Inference only.  `forward_clip` / `emotion2vec_features` process one clip (the un-contaminated equivalent of the
reference, which pads a batch with zeros and passes padding_mask=None, so shorter clips attend to their padding);
`forward_batch` runs equal-length clips in one pass and `emotion2vec_features_batch` reproduces the reference's
zero-padded batch semantics on top of it.  PARITY UNPINNED upstream: fairseq,
timm and the checkpoint are absent; tests pin the HIP path against a torch CPU restatement (oracle/emotion2vec_ref.py).
"""
import math

import torch
from torch import nn

from .. import ops
from .._lib import lib, check

CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2          # audio.py:25-28


def alibi_slopes(n):
    """get_alibi's head slopes (base.py:544-562)."""
    def pow2(k):
        start = 2 ** (-(2 ** -(math.log2(k) - 3)))
        return [start * start ** i for i in range(k)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][: n - c]


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class AltBlock(nn.Module):
    """Parameter holder of modules.py:184-230 (post-LN variant executed by Emotion2Vec.forward)."""

    def __init__(self, dim, heads, mlp_ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _BlockEncoder(nn.Module):
    def __init__(self, blocks, norm):
        super().__init__()
        self.blocks = blocks
        self.norm = norm


class _AudioEncoder(nn.Module):
    def __init__(self, embed_dim, prenet_depth, depth, heads, mlp_ratio, eps, conv_pos_depth, conv_pos_width,
                 conv_pos_groups, use_alibi, per_layer, per_head, alibi_scale):
        super().__init__()
        layers, cin = [], 1
        for dim, k, s in CONV_SPEC:
            layers.append(nn.Sequential(nn.Conv1d(cin, dim, k, stride=s, bias=False), nn.Dropout(0.0),
                                        nn.Sequential(nn.Identity(), nn.LayerNorm(dim), nn.Identity()), nn.GELU()))
            cin = dim
        self.local_encoder = nn.Module()
        self.local_encoder.conv_layers = nn.ModuleList(layers)
        self.project_features = nn.Sequential(nn.Identity(), nn.LayerNorm(cin), nn.Linear(cin, embed_dim))
        k = max(3, conv_pos_width // conv_pos_depth)
        self.relative_positional_encoder = nn.Sequential(
            nn.Identity(),
            *[nn.Sequential(nn.Conv1d(embed_dim, embed_dim, kernel_size=k, padding=k // 2, groups=conv_pos_groups),
                            nn.Identity(), nn.Identity(), nn.LayerNorm(embed_dim, elementwise_affine=False),
                            nn.Identity(), nn.GELU()) for _ in range(conv_pos_depth)],
            nn.Identity())
        self.context_encoder = _BlockEncoder(
            nn.ModuleList([AltBlock(embed_dim, heads, mlp_ratio, eps) for _ in range(prenet_depth)]),
            nn.LayerNorm(embed_dim, eps=eps))
        if use_alibi:
            self.alibi_scale = nn.Parameter(torch.full(((prenet_depth + depth) if per_layer else 1, 1,
                                                        heads if per_head else 1, 1, 1), float(alibi_scale)))
        else:
            self.alibi_scale = None


class Emotion2Vec(nn.Module):
    def __init__(self, embed_dim=768, depth=8, prenet_depth=4, num_heads=12, mlp_ratio=4.0, norm_eps=1e-6,
                 layer_norm_first=False, conv_pos_depth=5, conv_pos_width=95, conv_pos_groups=16,
                 use_alibi_encoder=True, learned_alibi_scale_per_layer=True, learned_alibi_scale_per_head=True,
                 alibi_scale=1.0):
        super().__init__()
        assert not layer_norm_first, "emotion2vec/data2vec-audio uses the post-LN blocks (config.py:40)"
        assert embed_dim % num_heads == 0 and (embed_dim // num_heads) % 4 == 0
        assert embed_dim // conv_pos_groups == 48 and max(3, conv_pos_width // conv_pos_depth) == 19, \
            "the grouped positional conv kernel is built for 48 channels/group, k=19 (the default config)"
        self.embed_dim, self.heads, self.prenet_depth, self.depth = embed_dim, num_heads, prenet_depth, depth
        self.conv_pos_groups = conv_pos_groups
        self.modality_encoders = nn.ModuleDict({"AUDIO": _AudioEncoder(
            embed_dim, prenet_depth, depth, num_heads, mlp_ratio, norm_eps, conv_pos_depth, conv_pos_width,
            conv_pos_groups, use_alibi_encoder, learned_alibi_scale_per_layer, learned_alibi_scale_per_head,
            alibi_scale)})
        self.blocks = nn.ModuleList([AltBlock(embed_dim, num_heads, mlp_ratio, norm_eps) for _ in range(depth)])
        self.register_buffer("_slopes", torch.tensor(alibi_slopes(num_heads), dtype=torch.float32), persistent=False)
        self.register_buffer("_ones", torch.ones(embed_dim), persistent=False)
        self.eval()

    # ---------------------------------------------------------------------------------------------
    @staticmethod
    def _ln(x, ln, act=0):
        return ops.layernorm(x, ln.weight, ln.bias, ln.eps, act)

    def _linear(self, x, lin, act=ops.ACT_NONE):
        return ops.linear(x, lin.weight, lin.bias, act)

    def _attention(self, x, attn, B, T, Tp, layer):
        """AltAttention.forward (modules.py:280-322) for B clips stacked as rows: x [B * Tp, D] (rows >= T of a clip are
        padding).  q / k / v projections over all rows, then ONE fused launch for softmax(q k^T * scale + alibi) v over all
        heads and clips (avdf_flash_attn_alibi: the [H, T, T] score matrices never reach HBM), then the projection."""
        D = x.shape[1]
        H, hd = self.heads, D // self.heads
        st = torch.cuda.current_stream().cuda_stream
        w, b = attn.qkv.weight, attn.qkv.bias
        q = ops.linear(x, w[0:D], b[0:D])
        k = ops.linear(x, w[D:2 * D], b[D:2 * D])
        v = ops.linear(x, w[2 * D:3 * D], b[2 * D:3 * D])
        out = torch.zeros((B * Tp, D), dtype=torch.float32, device=x.device)
        enc = self.modality_encoders["AUDIO"]
        if enc.alibi_scale is not None:
            a = enc.alibi_scale.clamp_min(0)
            sc = (a[layer] if a.shape[0] > 1 else a[0]).reshape(-1)       # [H] or [1]
            slopes = (self._slopes * sc).contiguous()
        else:
            slopes = torch.zeros(H, dtype=torch.float32, device=x.device)
        check(lib.avdf_flash_attn_alibi(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), slopes.data_ptr(), B, T, H, hd,
                                        D, Tp * D, hd ** -0.5, st), "flash_attn_alibi")
        return self._linear(out, attn.proj)

    def _block(self, x, blk, B, T, Tp, layer):
        """post-LN AltBlock (modules.py:242-249): x += attn(x); r = x = norm1(x); x = norm2(r + mlp(x))."""
        a = self._attention(x, blk.attn, B, T, Tp, layer)
        x = ops.affine_residual(x.unsqueeze(0), None, a.unsqueeze(0), None, self._ones)[0]
        r = self._ln(x, blk.norm1)
        m = self._linear(self._linear(r, blk.mlp.fc1, ops.ACT_GELU), blk.mlp.fc2)
        x = ops.affine_residual(r.unsqueeze(0), None, m.unsqueeze(0), None, self._ones)[0]
        return self._ln(x, blk.norm2)

    @torch.no_grad()
    def forward_clip(self, pcm):
        """pcm [N] fp32 @16 kHz on the device -> [(N - 400) // 320 + 1, 768]."""
        assert not self.training and pcm.dim() == 1 and pcm.is_cuda
        return self.forward_batch(pcm.unsqueeze(0))[0]

    @torch.no_grad()
    def forward_batch(self, pcm):
        """pcm [B, N] (equal-length clips, e.g. the zero-padded batch of the reference's collate) -> [B, T, 768].  Every
        row-wise operator (LayerNorm, the Linear layers: 8 of the 10 GEMMs of a block) runs ONCE over the B * T rows of
        the batch instead of per clip (M = 1535 -> B x 1536 rows per launch); the temporal operators (feature convs,
        positional convs, attention) keep the clips apart."""
        assert not self.training and pcm.dim() == 2 and pcm.is_cuda
        with ops.infer_mode():
            return self._forward_batch(pcm.contiguous())

    def _forward_batch(self, pcm):
        enc = self.modality_encoders["AUDIO"]
        st = torch.cuda.current_stream().cuda_stream
        dev = pcm.device
        B, n = pcm.shape
        layers = enc.local_encoder.conv_layers
        # --- local_features (base.py:189-202): conv stack -> LN -> Linear ---------------------------------
        dim0, k0, s0 = CONV_SPEC[0]
        T = (n - k0) // s0 + 1
        x = torch.empty((B, T, dim0), dtype=torch.float32, device=dev)
        w0 = layers[0][0].weight.contiguous()
        for b in range(B):
            check(lib.avdf_wave_conv_c1(pcm[b].data_ptr(), w0.data_ptr(), x[b].data_ptr(), n, T, dim0, k0, s0, st), "wave_conv_c1")
        x = self._ln(x.view(B * T, dim0), layers[0][2][1], 2)                # LayerNorm + GELU fused
        for i in range(1, len(CONV_SPEC)):
            dim, k, s = CONV_SPEC[i]
            Tn = (T - k) // s + 1
            y = torch.empty((B * Tn, dim), dtype=torch.float32, device=dev)
            cin = x.shape[1]
            check(lib.avdf_conv1d_fwd_ex(x.data_ptr(), layers[i][0].weight.contiguous().data_ptr(), None, y.data_ptr(),
                                         B, T, cin, Tn, dim, k, s, 0, cin, cin * k, dim, 0, st), "feature conv")
            x, T = self._ln(y, layers[i][2][1], 2), Tn
        x = self._linear(self._ln(x, enc.project_features[1]), enc.project_features[2])            # [B * T, D]
        # --- contextualized_features (base.py:204-329), mask=False ---------------------------------------
        Tp = (T + 3) // 4 * 4               # row padding keeps every GEMM operand stride 16-byte aligned
        D = self.embed_dim
        xp = torch.zeros((B, Tp, D), dtype=torch.float32, device=dev)
        xp[:, :T] = x.view(B, T, D)
        h = xp
        for i in range(1, len(enc.relative_positional_encoder) - 1):
            conv = enc.relative_positional_encoder[i][0]
            y = torch.zeros((B, Tp, D), dtype=torch.float32, device=dev)
            kpos = conv.weight.shape[2]
            # k=19 / 16 groups (every released checkpoint): each group is a 48-channel GEMM slice on the MFMA kernel
            # (2.15 GF per layer: 290 -> ~50 us); other kernel sizes keep the scalar kernel
            fn = lib.avdf_grouped_conv1d_fwd if (kpos == 19 and (D // self.conv_pos_groups) % 4 == 0) else lib.avdf_grouped_conv_pos
            wc, bc = conv.weight.contiguous(), conv.bias
            for b in range(B):
                check(fn(h[b].data_ptr(), wc.data_ptr(), bc.data_ptr(), y[b].data_ptr(), T, D, self.conv_pos_groups, kpos, st),
                      "grouped_conv_pos")
            h = ops.layernorm(y.view(B * Tp, D), None, None, 1e-5, 2).view(B, Tp, D)        # LN (no affine) + GELU
            if Tp > T:
                h[:, T:].zero_()                                             # the next conv's zero padding past the clip end
        x = ops.affine_residual(xp.view(1, B * Tp, D), None, h.view(1, B * Tp, D), None, self._ones)[0]
        x = self._ln(x, enc.context_encoder.norm)                            # BlockEncoder: norm first (modules.py:73-74)
        layer = 0
        for blk in enc.context_encoder.blocks:
            x = self._block(x, blk, B, T, Tp, layer)
            layer += 1
        for blk in self.blocks:
            x = self._block(x, blk, B, T, Tp, layer)
            layer += 1
        return x.view(B, Tp, D)[:, :T]

    @torch.no_grad()
    def extract_features(self, source, padding_mask=None, mask=False):
        """Reference entry point (emotion2vec.py:177-188): source [B, N] -> {'x': [B, T, 768]}."""
        assert padding_mask is None and not mask
        return {"x": self.forward_batch(source),
                "padding_mask": None, "layer_results": [], "mask": None}


@torch.no_grad()
def emotion2vec_features(pcm, model, n_samples=None, pad_samples=None):
    """PCM [N] -> the array the reference saves (scripts/extract_audio_feature_batch.py:85-98):
    feats['x'][: int(50 * len - 0.817)] with len in seconds.  pad_samples: waveform length the clip is zero-padded to
    before the encoder (see `emotion2vec_features_batch`)."""
    n = pcm.numel() if n_samples is None else n_samples
    if pad_samples is not None and pad_samples > pcm.numel():
        pcm = torch.nn.functional.pad(pcm, (0, pad_samples - pcm.numel()))
    x = model.forward_clip(pcm)
    return x[:int(50 * (n / 16000.0) - 0.817)]


@torch.no_grad()
def emotion2vec_features_equal_length(pcm, model):
    """pcm [B, N] of equal-length clips -> list of per-clip feature arrays, one batched encoder pass."""
    x = model.forward_batch(pcm)
    k = int(50 * (pcm.shape[1] / 16000.0) - 0.817)
    return [x[b, :k] for b in range(pcm.shape[0])]


@torch.no_grad()
def emotion2vec_features_batch(pcms, model):
    """The reference's batch semantics (scripts/extract_audio_feature_batch.py:35-52,73-91): waveforms zero-padded to the
    longest of the batch, `extract_features(wav, padding_mask=None)` -- so the attention of a shorter clip also attends to
    its padding positions -- then rows [: int(50 * len - 0.817)] per clip.  No operator of the encoder mixes clips, so
    clip i of the padded batch is the per-clip forward of its zero-padded waveform."""
    longest = max(int(p.numel()) for p in pcms)
    batch = torch.stack([torch.nn.functional.pad(p, (0, longest - p.numel())) for p in pcms])
    x = model.forward_batch(batch)
    return [x[i, :int(50 * (p.numel() / 16000.0) - 0.817)] for i, p in enumerate(pcms)]
