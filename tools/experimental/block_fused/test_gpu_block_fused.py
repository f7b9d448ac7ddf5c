"""Row-block fused launches of one attention half-block (csrc/block_fused.hip: avdf_qkv_fused_fwd,
avdf_attn_out_fused_fwd) against a float64 torch statement of the reference block
(libs/modeling/blocks.py:1152-1224 attention, 1307-1313 / 866-872 pre-norms, residual, LN2): forward values, every
input / parameter gradient (the backward is the per-op one, fed by the tensors the fused launches save), the
accumulate-into-main_grad engine path, inference (nothing but the outputs is written), strides, shared / distinct
inputs and the folded nearest resampling of the cross-level blocks.

Runs against a tree with `integration.patch` applied and the library rebuilt (see README.md here):
    python -m pytest tools/experimental/block_fused/test_gpu_block_fused.py"""
import math
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))   # repo root

pytestmark = pytest.mark.gpu

C = 256


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def g32(t, grad=False):
    return t.float().to("cuda:0").requires_grad_(grad)


def close(got, ref, what, rtol=2e-5, floor=1e-6):
    """floor: gradients that are mathematically zero (key bias and key-norm beta under the softmax: a constant added to every
    key shifts all scores of a query alike) are pure fp32 rounding noise of O(1) summands -- held to rtol * floor."""
    got, ref = got.detach().cpu().double(), ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    scale = max(float(ref.abs().max()), floor)
    err = float((got - ref).abs().max()) / scale
    assert err <= rtol, f"{what}: rel err {err:.3e} > {rtol:.1e} (scale {scale:.3e})"


def banded_attention(q, k, v, mask, H, W, rel=None, tw=None):
    """LocalMaskedMHCA's core (blocks.py:1180-1216) stated naively: scores over the 2w+1 band, -1e4 on masked keys, softmax,
    masked query rows zeroed, optional per-head bias before / weight after the softmax."""
    B, T, Cc = q.shape
    hs, w = Cc // H, W // 2
    qh, kh, vh = (t.view(B, T, H, hs) for t in (q, k, v))
    idx = torch.arange(T)[:, None] + torch.arange(W)[None, :]                   # rows of the zero-padded key axis
    kp, vp = F.pad(kh, (0, 0, 0, 0, w, w)), F.pad(vh, (0, 0, 0, 0, w, w))
    s = torch.einsum("bthd,btjhd->bthj", qh, kp[:, idx]) / math.sqrt(hs)
    if rel is not None:
        s = s + rel.view(1, 1, H, W)
    mk = F.pad(mask, (w, w))[:, idx]                                            # [B, T, W]
    s = s + torch.where(mk == 0, -1e4, 0.0)[:, :, None, :]
    valid = ((idx - w) >= 0) & ((idx - w) < T)
    s = s.masked_fill(~valid[None, :, None, :], float("-inf"))
    p = torch.softmax(s, -1) * mask[:, :, None, None]
    if tw is not None:
        p = p * tw.view(1, 1, H, W)
    return torch.einsum("bthj,btjhd->bthd", p, vp[:, idx]).reshape(B, T, Cc)


def resample(x, sh):
    return x if sh == 0 else (x.repeat_interleave(1 << sh, dim=1) if sh > 0 else x[:, ::(1 << -sh)])


CASES = [
    # B, Tq_in, stride, share, kv_shift, H, W, use_keep, tables
    (2, 96, 1, "all", 0, 4, 7, True, False),
    (8, 768, 1, "all", 0, 4, 7, True, False),        # the top pyramid level of the headline step
    (3, 96, 2, "all", 0, 4, 7, False, False),        # strided block: Tout = 48 = one full + one half tile; max-pooled skip
    (2, 96, 1, "kv", 1, 4, 7, True, False),          # lh_branch: k / v from the level below, read through the row map
    (2, 48, 1, "kv", -1, 8, 9, False, True),         # hh_branch: k / v from the level above; 8 heads, window 9, rel_pe + time weights
    (1, 40, 1, "none", 0, 4, 5, False, False),       # three distinct inputs / pre-norms, window 5
]


def build(case, seed0=0):
    B, T, stride, share, sh, H, W, use_keep, tables = case
    Tout = T // stride
    Tsrc = T if sh == 0 else ((T >> sh) if sh > 0 else (T << -sh))
    xq = rnd(B, T, C, seed=seed0 + 1)
    xs = {"all": [xq, xq, xq], "kv": None, "none": [xq, rnd(B, T, C, seed=seed0 + 2), rnd(B, T, C, seed=seed0 + 3)]}[share]
    if xs is None:
        xk = rnd(B, Tsrc, C, seed=seed0 + 2)
        xs = [xq, xk, xk]
    npar = 1 if share == "all" else 3
    la = [(1 + 0.2 * rnd(1, C, 1, seed=10 + i), 0.2 * rnd(1, C, 1, seed=20 + i)) for i in range(npar)]
    la = [la[0]] * 3 if npar == 1 else la
    dw = [rnd(C, 1, 3, seed=30 + i, scale=0.5) for i in range(3)]
    lb = [(1 + 0.2 * rnd(1, C, 1, seed=40 + i), 0.2 * rnd(1, C, 1, seed=50 + i)) for i in range(3)]
    pw = [rnd(C, C, 1, seed=60 + i, scale=C ** -0.5) for i in range(4)]          # q, k, v, out projections
    pb = [rnd(C, seed=70 + i, scale=0.1) for i in range(4)]
    lens = [Tout - 5 * b for b in range(B)]
    mask = (torch.arange(Tout)[None, :] < torch.tensor(lens)[:, None]).double()
    scale = rnd(1, C, 1, seed=80, scale=0.3)
    keep = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.9, 0.0, 1.0 / 0.9, 1.0 / 0.9, 1.0 / 0.9][:B]).double() if use_keep else None
    g2, b2 = 1 + 0.2 * rnd(1, C, 1, seed=81), 0.2 * rnd(1, C, 1, seed=82)
    rel = rnd(1, 1, H, W, seed=83, scale=0.5) if tables else None
    tw = (1 + 0.3 * rnd(1, H, W, seed=84)) if tables else None
    return dict(xs=xs, la=la, dw=dw, lb=lb, pw=pw, pb=pb, mask=mask, scale=scale, keep=keep, g2=g2, b2=b2, rel=rel, tw=tw,
                Tout=Tout)


def reference(case, d, leaf):
    B, T, stride, share, sh, H, W, use_keep, tables = case
    eps = 1e-5
    shifts = (0, sh, sh)
    qkv = []
    for i in range(3):
        x = resample(leaf(d["xs"][i]), shifts[i])
        z = F.layer_norm(x, (C,), leaf(d["la"][i][0]).view(-1), leaf(d["la"][i][1]).view(-1), eps)
        u = F.conv1d(z.transpose(1, 2), leaf(d["dw"][i]), None, stride, 1, groups=C).transpose(1, 2) * d["mask"][..., None]
        y = F.layer_norm(u, (C,), leaf(d["lb"][i][0]).view(-1), leaf(d["lb"][i][1]).view(-1), eps)
        qkv.append(F.conv1d(y.transpose(1, 2), leaf(d["pw"][i]), leaf(d["pb"][i])).transpose(1, 2))
    att = banded_attention(*qkv, d["mask"], H, W, None if d["rel"] is None else leaf(d["rel"]),
                           None if d["tw"] is None else leaf(d["tw"]))
    r = F.conv1d(att.transpose(1, 2), leaf(d["pw"][3]), leaf(d["pb"][3])).transpose(1, 2) * d["mask"][..., None]
    xq = leaf(d["xs"][0])
    skip = xq if stride == 1 else F.max_pool1d(xq.transpose(1, 2), 3, 2, 1).transpose(1, 2)
    fr = 1.0 if d["keep"] is None else d["keep"].view(B, 1, 1)
    out = skip * d["mask"][..., None] + leaf(d["scale"]).view(1, 1, C) * fr * r
    h = F.layer_norm(out, (C,), leaf(d["g2"]).view(-1), leaf(d["b2"]).view(-1), eps)
    return qkv, out, h


def run_fused(ops, case, d, dleaf, grad):
    B, T, stride, share, sh, H, W, use_keep, tables = case
    xs = [dleaf(x) for x in d["xs"]]
    res = ops.qkv_fused(xs, [(dleaf(a), dleaf(b)) for a, b in d["la"]], [dleaf(w) for w in d["dw"]],
                        [(dleaf(a), dleaf(b)) for a, b in d["lb"]], [dleaf(w) for w in d["pw"][:3]],
                        [dleaf(b) for b in d["pb"][:3]], g32(d["mask"]), stride, 1e-5, passthrough=grad,
                        xshift=(0, sh, sh) if sh else None)
    q, k, v = res[:3]
    x_res = res[3] if grad else xs[0]
    skip = x_res if stride == 1 else ops.maxpool3s2(x_res)
    out, h = ops.attn_out_fused(q, k, v, g32(d["mask"]), H, W, None if d["rel"] is None else dleaf(d["rel"]),
                                None if d["tw"] is None else dleaf(d["tw"]), dleaf(d["pw"][3]), dleaf(d["pb"][3]), skip,
                                dleaf(d["scale"]), None if d["keep"] is None else g32(d["keep"]), dleaf(d["g2"]), dleaf(d["b2"]),
                                1e-5)
    return (q, k, v), out, h


@pytest.mark.parametrize("main", [False, True])
@pytest.mark.parametrize("case", CASES)
def test_fused_half_block_matches_float64(case, main):
    from audio_visual_deepfake_detection_amd import ops
    assert ops.FUSE_BLOCK
    d = build(case)
    uniq = {}

    def leaf(t):
        if id(t) not in uniq:
            uniq[id(t)] = t.clone().requires_grad_(True)
        return uniq[id(t)]

    qkv_ref, out_ref, h_ref = reference(case, d, leaf)
    d_out, d_h = rnd(*out_ref.shape, seed=90), rnd(*h_ref.shape, seed=91)
    ((out_ref * d_out).sum() + (h_ref * d_h).sum()).backward()

    dmap, base = {}, {}

    def dleaf(t):
        if id(t) not in dmap:
            dmap[id(t)] = g32(t, True)
        return dmap[id(t)]

    if main:          # engine path: every parameter accumulates into a flat-buffer view on top of what is there
        params = [a for pr in d["la"] for a in pr] + d["dw"] + [a for pr in d["lb"] for a in pr] + d["pw"] + d["pb"] + \
                 [d["scale"], d["g2"], d["b2"]] + ([d["rel"], d["tw"]] if d["rel"] is not None else [])
        for t in params:
            p = dleaf(t)
            if not hasattr(p, "main_grad"):
                p.main_grad = g32(rnd(*t.shape, seed=100 + len(base)))
                base[id(t)] = p.main_grad.clone()
    qkv, out, h = run_fused(ops, case, d, dleaf, True)
    for i, nm in enumerate("qkv"):
        close(qkv[i], qkv_ref[i], f"{nm} projection")
    close(out, out_ref, "residual stream after the attention half")
    close(h, h_ref, "LN2 output")
    ((out * g32(d_out)).sum() + (h * g32(d_h)).sum()).backward()
    torch.cuda.synchronize()
    for key, t64 in uniq.items():
        what = f"gradient of tensor shape {tuple(t64.shape)}"
        if key in base:
            assert dmap[key].grad is None, "accumulate path must not hand a parameter gradient to autograd"
            close(dmap[key].main_grad - base[key], t64.grad, "main_grad: " + what, rtol=1e-4, floor=0.5)
        else:
            close(dmap[key].grad, t64.grad, what, rtol=1e-4, floor=0.5)


@pytest.mark.parametrize("case", CASES)
def test_fused_half_block_inference_equals_training_forward(case):
    """no_grad: the launches get NULL save-tables and must produce the same outputs bit for bit."""
    from audio_visual_deepfake_detection_amd import ops
    d = build(case, seed0=200)
    dmap = {}

    def dleaf(t):
        if id(t) not in dmap:
            dmap[id(t)] = g32(t, True)
        return dmap[id(t)]

    qkv_t, out_t, h_t = run_fused(ops, case, d, dleaf, True)
    with torch.no_grad():
        qkv_i, out_i, h_i = run_fused(ops, case, d, dleaf, False)
    torch.cuda.synchronize()
    assert out_i.grad_fn is None and out_t.grad_fn is not None
    for a, b in zip(qkv_t + (out_t, h_t), qkv_i + (out_i, h_i)):
        assert torch.equal(a.detach(), b)


def test_fused_blocks_equal_per_op_blocks_in_a_backbone_forward_backward():
    """The block modules pick the fused launches by themselves (C = 256): a stride-1 TransformerBlock, a strided one and a
    cross-level MutilModelTransformerBlock give the per-op path's outputs and gradients (AVDF_FUSE_BLOCK=0 switch)."""
    from audio_visual_deepfake_detection_amd import ops
    from audio_visual_deepfake_detection_amd.libs.modeling.blocks import TransformerBlock, MutilModelTransformerBlock
    torch.manual_seed(5)
    dev = "cuda:0"
    blocks = [TransformerBlock(C, 4, (1, 1), mha_win_size=7, path_pdrop=0.1).to(dev),
              TransformerBlock(C, 4, (2, 2), mha_win_size=7, path_pdrop=0.1).to(dev),
              MutilModelTransformerBlock(C, 4, (1, 1), mha_win_size=7, path_pdrop=0.1).to(dev)]
    for blk in blocks:
        blk.train()
        for m in blk.modules():
            if hasattr(m, "drop_prob"):
                m.drop_prob = 0.0                       # deterministic comparison; keep factors are covered above
        with torch.no_grad():
            blk.drop_path_attn.scale.normal_(0, 0.3)
            blk.drop_path_mlp.scale.normal_(0, 0.3)
    B, T = 3, 96
    x0 = torch.randn(B, T, C, device=dev)
    mask = (torch.arange(T, device=dev)[None, :] < torch.tensor([96, 80, 50], device=dev)[:, None]).float()

    def run(fuse):
        ops.FUSE_BLOCK = fuse
        try:
            for blk in blocks:
                blk.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            y1, m1 = blocks[0](x, mask)
            y2, m2 = blocks[1](y1, m1)                                       # [B, 48, C]
            y3, _ = blocks[2](y1, m1, x_k=y2, mask_k=m1, x_v=y2, mask_v=m1, kv_shift=1)
            (y3 * torch.linspace(-1, 1, C, device=dev)).sum().backward()
            grads = [p.grad.clone() for blk in blocks for p in blk.parameters() if p.grad is not None]
            return y3.detach().clone(), x.grad.clone(), grads
        finally:
            ops.FUSE_BLOCK = True

    ya, gxa, ga = run(True)
    yb, gxb, gb = run(False)
    assert len(ga) == len(gb) and len(ga) > 30
    close(ya, yb.cpu(), "block stack output", rtol=1e-5)
    close(gxa, gxb.cpu(), "block stack input gradient", rtol=1e-4)
    for a, b in zip(ga, gb):
        close(a, b.cpu(), f"parameter gradient {tuple(a.shape)}", rtol=1e-4, floor=0.5)
