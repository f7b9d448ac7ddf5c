"""TEST INFRASTRUCTURE ONLY -- torch CPU restatement of the Emotion2Vec inference path.

PARITY UNPINNED: fairseq / timm / omegaconf and emotion2vec_base.pt are absent, so this restates
audio_feature/emotion_audio/upstream/models/{emotion2vec.py:97-188, audio.py:52-138, base.py:189-329,538-642,
modules.py:63-97,184-322} and fairseq's ConvFeatureExtractionModel(mode='layer_norm', conv_bias=False) from the
cited sources + PyTorch primitive semantics.  It pins HIP == torch, not HIP == fairseq."""
import math

import torch
import torch.nn.functional as F

CONV_SPEC = [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2


def slopes(n):
    def pow2(k):
        start = 2 ** (-(2 ** -(math.log2(k) - 3)))
        return [start * start ** i for i in range(k)]
    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + slopes(2 * c)[0::2][: n - c]


def alibi_bias(T, heads):
    """get_alibi(T, heads) (upstream/models/base.py:538-604, dims=1): slopes[h] * -|i - j| -> [H, T, T].
    Pinned: equals the reference function exactly (tests/golden/emotion2vec_parts.npz)."""
    pos = -(torch.arange(T)[None, :] - torch.arange(T)[:, None]).abs()
    return torch.Tensor(slopes(heads))[:, None, None] * pos[None]


def alt_attention(sd, p, x, heads, bias):
    """AltAttention.forward (upstream/models/modules.py:280-322; no cosine attention, no padding mask): x [B, N, C],
    bias [H, N, N] or None.  Pinned to the reference class (tests/golden/emotion2vec_parts.npz)."""
    B, N, C = x.shape
    qkv = F.linear(x, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(B, N, 3, heads, C // heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    attn = (q * (C // heads) ** -0.5) @ k.transpose(-2, -1)
    if bias is not None:
        attn = attn + bias
    attn = attn.softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])


def alt_block(sd, p, x, heads, bias, eps):
    B, N, C = x.shape
    x = x + alt_attention(sd, p, x, heads, bias)
    r = x = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
    x = F.linear(F.gelu(F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"],
                 sd[p + "mlp.fc2.bias"])
    return F.layer_norm(r + x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)


def forward(sd, pcm, heads=12, prenet_depth=4, depth=8, eps=1e-6, conv_pos_depth=5):
    """pcm [B, N] -> [B, T, 768] (extract_features, mask=False, padding_mask=None)."""
    a = "modality_encoders.AUDIO."
    x = pcm.unsqueeze(1)
    for i, (dim, k, s) in enumerate(CONV_SPEC):
        x = F.conv1d(x, sd[f"{a}local_encoder.conv_layers.{i}.0.weight"], None, stride=s)
        x = F.layer_norm(x.transpose(1, 2), (dim,), sd[f"{a}local_encoder.conv_layers.{i}.2.1.weight"],
                         sd[f"{a}local_encoder.conv_layers.{i}.2.1.bias"], 1e-5).transpose(1, 2)
        x = F.gelu(x)
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd[a + "project_features.1.weight"], sd[a + "project_features.1.bias"], 1e-5)
    x = F.linear(x, sd[a + "project_features.2.weight"], sd[a + "project_features.2.bias"])
    h = x.transpose(1, 2)
    for i in range(1, conv_pos_depth + 1):
        w = sd[f"{a}relative_positional_encoder.{i}.0.weight"]
        h = F.conv1d(h, w, sd[f"{a}relative_positional_encoder.{i}.0.bias"], padding=w.shape[-1] // 2, groups=16)
        h = F.gelu(F.layer_norm(h.transpose(1, 2), (h.shape[1],), None, None, 1e-5).transpose(1, 2))
    x = x + h.transpose(1, 2)
    B, T, C = x.shape
    bias_base, scale = None, None
    if a + "alibi_scale" in sd:
        bias_base = alibi_bias(T, heads)                                            # [H, T, T]
        scale = sd[a + "alibi_scale"].clamp_min(0)
        if scale.shape[0] == 1:
            bias_base = bias_base * scale[0, 0]
            scale = None
    x = F.layer_norm(x, (C,), sd[a + "context_encoder.norm.weight"], sd[a + "context_encoder.norm.bias"], eps)
    layer = 0
    for i in range(prenet_depth):
        b = None if bias_base is None else (bias_base if scale is None else bias_base * scale[layer, 0])
        x = alt_block(sd, f"{a}context_encoder.blocks.{i}.", x, heads, b, eps)
        layer += 1
    for i in range(depth):
        b = None if bias_base is None else (bias_base if scale is None else bias_base * scale[layer, 0])
        x = alt_block(sd, f"blocks.{i}.", x, heads, b, eps)
        layer += 1
    return x
