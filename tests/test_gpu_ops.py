"""Per-kernel parity on a real MI355X: every C-ABI op (forward and backward) against a plain
PyTorch float64 CPU statement of the same reference op.  fp32 kernels, so the tolerance is a few
fp32 ulps of the result scale (1e-5 relative to max|ref|, far inside the 1e-3 of the north star)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def dev():
    return torch.device("cuda:0")


def close(got, ref, what, rtol=RTOL):
    got = got.detach().cpu().double()
    ref = ref.detach().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    scale = max(float(ref.abs().max()), 1e-6)
    err = float((got - ref).abs().max()) / scale
    assert err <= rtol, f"{what}: rel err {err:.3e} > {rtol:.1e} (scale {scale:.3e})"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def mk_mask(B, T, lens):
    return (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).double()


def g32(t, grad=False):
    return t.float().to(dev()).requires_grad_(grad)


@pytest.fixture(scope="module")
def ops():
    from audio_visual_deepfake_detection_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, T, Cin, Cout, K, stride, bias, mask, act
    (2, 96, 40, 32, 3, 1, True, True, 0),
    (2, 200, 96, 520, 3, 1, False, True, 1),
    (3, 96, 64, 36, 3, 2, True, True, 0),
    (2, 72, 256, 256, 1, 1, True, False, 2),
    (2, 50, 32, 128, 1, 1, False, True, 0),
    (6, 768, 64, 256, 3, 1, False, True, 0),      # 128-row tile path
    (6, 768, 64, 64, 3, 2, True, True, 0),
    (6, 768, 256, 128, 1, 1, True, False, 2),
    (1, 24, 2048, 256, 3, 2, True, True, 0),      # deep K loop, tiny T
    (4, 1, 512, 256, 1, 1, False, False, 0),      # Linear on [B, 2H]
    (4, 192, 512, 1024, 3, 2, True, True, 0),     # contraction down_3 at full size: big-weight wgrad, 2 token tiles per clip
    (4, 384, 256, 512, 3, 2, True, True, 0),      # contraction down_2
    (8, 768, 256, 1024, 1, 1, True, False, 2),    # fc1 at full size
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d(ops, case):
    B, T, Cin, Cout, K, s, has_bias, has_mask, act = case
    x = rnd(B, T, Cin, seed=1)
    w = rnd(Cout, Cin, K, seed=2, scale=1.0 / (Cin * K) ** 0.5)
    b = rnd(Cout, seed=3) if has_bias else None
    Tout = (T - 1) // s + 1 if K == 3 else T
    m = mk_mask(B, Tout, [Tout - (i * 7) % max(Tout // 2, 1) for i in range(B)]) if has_mask else None
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if has_bias else None
    y = F.conv1d(xr.transpose(1, 2), wr, br, stride=s, padding=K // 2).transpose(1, 2)
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.gelu(y)
    if m is not None:
        y = y * m[:, :, None]
    gy = rnd(*y.shape, seed=4)
    y.backward(gy)
    xg, wg = g32(x, True), g32(w, True)
    bg = g32(b, True) if has_bias else None
    yo = ops.conv1d(xg, wg, bg, g32(m) if has_mask else None, s, act)
    close(yo, y, "conv fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "conv dgrad")
    close(wg.grad, wr.grad, "conv wgrad")
    if has_bias:
        close(bg.grad, br.grad, "conv dbias")


@pytest.mark.parametrize("C,relu,use_pe", [(32, False, False), (256, True, True), (1024, True, False), (516, False, False)])
def test_layernorm(ops, C, relu, use_pe):
    B, T = 3, 50
    x = rnd(B, T, C, seed=5, scale=2.0) + 0.5
    g, b = rnd(1, C, 1, seed=6) * 0.2 + 1.0, rnd(1, C, 1, seed=7) * 0.2
    pe = rnd(T, C, seed=8) if use_pe else None
    m = mk_mask(B, T, [50, 31, 7])
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gr.view(-1), br.view(-1), 1e-5)
    if relu:
        y = F.relu(y)
    if use_pe:
        y = y + pe[None] * m[:, :, None]
    gy = rnd(B, T, C, seed=9)
    y.backward(gy)
    xg, gg, bg = g32(x, True), g32(g, True), g32(b, True)
    yo = ops.layernorm(xg, gg, bg, 1e-5, relu, g32(pe) if use_pe else None, g32(m) if use_pe else None)
    close(yo, y, "ln fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "ln dx")
    close(gg.grad, gr.grad, "ln dgamma")
    close(bg.grad, br.grad, "ln dbeta")


@pytest.mark.parametrize("C,relu", [(256, True), (32, False), (516, True)])
def test_layernorm_with_row_scale(ops, C, relu):
    """y = act(LN(x)) * rowscale[row] in ONE launch (the heads' LayerNorms on the gapped level sequence), forward and
    backward, against the composition of the plain ops in float64."""
    B, T = 3, 50
    x = rnd(B, T, C, seed=5, scale=2.0) + 0.5
    g, b = rnd(1, C, 1, seed=6) * 0.2 + 1.0, rnd(1, C, 1, seed=7) * 0.2
    rs = (torch.rand(B, T, generator=torch.Generator().manual_seed(3)) > 0.3).double() * (1.0 + 0.5 * rnd(B, T, seed=4).abs())
    xr, gr, br = x.clone().requires_grad_(True), g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gr.view(-1), br.view(-1), 1e-5)
    if relu:
        y = F.relu(y)
    y = y * rs[:, :, None]
    gy = rnd(B, T, C, seed=9)
    y.backward(gy)
    xg, gg, bg = g32(x, True), g32(g, True), g32(b, True)
    yo = ops.layernorm(xg, gg, bg, 1e-5, relu, None, None, g32(rs))
    close(yo, y, "ln*rowscale fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "ln*rowscale dx")
    close(gg.grad, gr.grad, "ln*rowscale dgamma")
    close(bg.grad, br.grad, "ln*rowscale dbeta")


@pytest.mark.parametrize("stride,C,T", [(1, 32, 96), (2, 256, 100), (1, 260, 10)])
def test_dwconv(ops, stride, C, T):
    B = 3
    x, w = rnd(B, T, C, seed=1), rnd(C, 1, 3, seed=2)
    Tout = (T - 1) // stride + 1
    m = mk_mask(B, Tout, [Tout, Tout // 2 + 1, 1])
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv1d(xr.transpose(1, 2), wr, None, stride=stride, padding=1, groups=C).transpose(1, 2) * m[:, :, None]
    gy = rnd(*y.shape, seed=3)
    y.backward(gy)
    xg, wg = g32(x, True), g32(w, True)
    yo = ops.dwconv(xg, wg, g32(m), stride)
    close(yo, y, "dwconv fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "dwconv dx")
    close(wg.grad, wr.grad, "dwconv dw")


def ref_local_attn(q, k, v, m, H, W):
    B, T, C = q.shape
    hs = C // H
    w = W // 2
    qh = q.view(B, T, H, hs).permute(0, 2, 1, 3) / hs ** 0.5
    kh = k.view(B, T, H, hs).permute(0, 2, 1, 3)
    vh = v.view(B, T, H, hs).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2)
    idx = torch.arange(T)
    band = (idx[None, :] - idx[:, None]).abs() <= w
    s = s + ((1 - m) * -1e4)[:, None, None, :]
    s = s.masked_fill(~band[None, None], float("-inf"))
    p = torch.softmax(s, -1) * m[:, None, :, None]
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, T, C)


@pytest.mark.parametrize("H,hs,W,T", [(4, 8, 7, 48), (4, 64, 7, 96), (2, 16, 3, 24), (4, 32, 9, 64)])
def test_local_attention(ops, H, hs, W, T):
    B, C = 2, H * hs
    q, k, v = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2), rnd(B, T, C, seed=3)
    m = mk_mask(B, T, [T, T // 2 + 3])
    qr, kr, vr = [t.clone().requires_grad_(True) for t in (q, k, v)]
    y = ref_local_attn(qr, kr, vr, m, H, W)
    gy = rnd(B, T, C, seed=4)
    y.backward(gy)
    qg, kg, vg = g32(q, True), g32(k, True), g32(v, True)
    yo = ops.local_attention(qg, kg, vg, g32(m), H, W)
    close(yo, y, "local attn fwd")
    yo.backward(g32(gy))
    close(qg.grad, qr.grad, "local attn dq")
    close(kg.grad, kr.grad, "local attn dk")
    close(vg.grad, vr.grad, "local attn dv")


@pytest.mark.parametrize("H,hs,Tq,Tk", [(4, 64, 24, 24), (2, 16, 6, 6), (4, 8, 70, 130)])
def test_global_attention(ops, H, hs, Tq, Tk):
    B, C = 2, H * hs
    q, k, v = rnd(B, Tq, C, seed=1), rnd(B, Tk, C, seed=2), rnd(B, Tk, C, seed=3)
    m = mk_mask(B, Tk, [Tk, Tk // 2 + 1])
    qr, kr, vr = [t.clone().requires_grad_(True) for t in (q, k, v)]
    qh = qr.view(B, Tq, H, hs).permute(0, 2, 1, 3) / hs ** 0.5
    kh = kr.view(B, Tk, H, hs).permute(0, 2, 1, 3)
    vh = vr.view(B, Tk, H, hs).permute(0, 2, 1, 3)
    att = (qh @ kh.transpose(-1, -2)).masked_fill(m[:, None, None, :] == 0, float("-inf"))
    y = (torch.softmax(att, -1) @ (vh * m[:, None, :, None])).permute(0, 2, 1, 3).reshape(B, Tq, C)
    gy = rnd(B, Tq, C, seed=4)
    y.backward(gy)
    qg, kg, vg = g32(q, True), g32(k, True), g32(v, True)
    yo = ops.global_attention(qg, kg, vg, g32(m), H)
    close(yo, y, "global attn fwd")
    yo.backward(g32(gy))
    close(qg.grad, qr.grad, "global attn dq")
    close(kg.grad, kr.grad, "global attn dk")
    close(vg.grad, vr.grad, "global attn dv")


def test_maxpool_and_residual(ops):
    B, T, C = 3, 96, 64
    x = rnd(B, T, C, seed=1)
    xr = x.clone().requires_grad_(True)
    y = F.max_pool1d(xr.transpose(1, 2), 3, 2, 1).transpose(1, 2)
    gy = rnd(*y.shape, seed=2)
    y.backward(gy)
    xg = g32(x, True)
    yo = ops.maxpool3s2(xg)
    close(yo, y, "maxpool fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "maxpool bwd")

    a, r = rnd(B, T, C, seed=3), rnd(B, T, C, seed=4)
    sc = rnd(1, C, 1, seed=5)
    keep = torch.tensor([1.0 / 0.9, 0.0, 1.0 / 0.9], dtype=torch.float64)
    ma, mr = mk_mask(B, T, [96, 40, 9]), mk_mask(B, T, [90, 41, 96])
    ar, rr, sr = a.clone().requires_grad_(True), r.clone().requires_grad_(True), sc.clone().requires_grad_(True)
    out = ar * ma[:, :, None] + sr.view(1, 1, C) * keep[:, None, None] * (rr * mr[:, :, None])
    go = rnd(B, T, C, seed=6)
    out.backward(go)
    ag, rg, sg = g32(a, True), g32(r, True), g32(sc, True)
    oo = ops.affine_residual(ag, g32(ma), rg, g32(mr), sg, g32(keep))
    close(oo, out, "affine_residual fwd")
    oo.backward(g32(go))
    close(ag.grad, ar.grad, "affine_residual da")
    close(rg.grad, rr.grad, "affine_residual dr")
    close(sg.grad, sr.grad, "affine_residual dscale")
    # optional operands absent
    o2 = ops.affine_residual(g32(a), None, g32(r), None, g32(sc), None)
    close(o2, a + sc.view(1, 1, C) * r, "affine_residual (no masks)")


def test_resample_transpose(ops):
    B, T, C = 2, 24, 32
    x = rnd(B, T, C, seed=1)
    for r in (2, 4, 32):
        xr = x.clone().requires_grad_(True)
        y = xr.repeat_interleave(r, dim=1)
        gy = rnd(*y.shape, seed=2)
        y.backward(gy)
        xg = g32(x, True)
        yo = ops.upsample(xg, r)
        close(yo, y, "upsample fwd")
        yo.backward(g32(gy))
        close(xg.grad, xr.grad, "upsample bwd")
    big = rnd(B, 96, C, seed=3)
    for r in (2, 8):
        br = big.clone().requires_grad_(True)
        y = br[:, ::r]
        gy = rnd(*y.shape, seed=4)
        y.backward(gy)
        bg = g32(big, True)
        yo = ops.downsample(bg, r)
        close(yo, y, "downsample fwd")
        yo.backward(g32(gy))
        close(bg.grad, br.grad, "downsample bwd")
    a, s = rnd(B, 48, C, seed=5), rnd(B, 24, C, seed=6)
    ar, sr = a.clone().requires_grad_(True), s.clone().requires_grad_(True)
    out = ar + sr.repeat_interleave(2, dim=1)
    go = rnd(B, 48, C, seed=7)
    out.backward(go)
    ag, sg = g32(a, True), g32(s, True)
    oo = ops.add_upsampled(ag, sg, 2)
    close(oo, out, "add_upsampled fwd")
    oo.backward(g32(go))
    close(ag.grad, ar.grad, "add_upsampled da")
    close(sg.grad, sr.grad, "add_upsampled dsmall")
    z = rnd(3, 45, 70, seed=8)
    close(ops.transpose_last2(g32(z)), z.float().transpose(1, 2), "transpose", rtol=0.0)
    rows = rnd(12, seed=9)
    close(ops.scale_rows(g32(z[0, :12]), g32(rows)), z[0, :12] * rows[:, None], "scale_rows")


@pytest.mark.parametrize("B,T,C", [(2, 384, 256), (3, 24, 100), (2, 48, 2048)])
def test_instnorm_pool(ops, B, T, C):
    x = rnd(B, T, C, seed=1, scale=1.5) + 0.3
    xr = x.clone().requires_grad_(True)
    y = F.leaky_relu(F.instance_norm(xr.transpose(1, 2), eps=1e-5), 0.2).transpose(1, 2)
    gy = rnd(B, T, C, seed=2)
    y.backward(gy)
    xg = g32(x, True)
    yo = ops.instnorm_lrelu(xg, 1e-5, 0.2)
    close(yo, y, "instnorm fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "instnorm bwd", rtol=1e-4)
    xr2 = x.clone().requires_grad_(True)
    p = torch.cat((xr2.max(dim=1).values, xr2.mean(dim=1)), dim=1)
    gp = rnd(B, 2 * C, seed=3)
    p.backward(gp)
    xg2 = g32(x, True)
    po = ops.maxavgpool(xg2)
    close(po, p, "maxavgpool fwd")
    po.backward(g32(gp))
    close(xg2.grad, xr2.grad, "maxavgpool bwd")


@pytest.mark.parametrize("N,K,Cin,T,use_scale", [(1, 3, 256, 96, False), (2, 3, 256, 50, True), (1, 1, 32, 1, False),
                                                 (2, 3, 520, 24, True)])
def test_smallconv(ops, N, K, Cin, T, use_scale):
    B = 3
    x = rnd(B, T, Cin, seed=1)
    w = rnd(N, Cin, K, seed=2, scale=1.0 / (Cin * K) ** 0.5)
    b = rnd(N, seed=3)
    m = mk_mask(B, T, [T, max(T // 2, 1), 1])
    sc = torch.tensor(1.3, dtype=torch.float64)
    xr, wr, br, sr = [t.clone().requires_grad_(True) for t in (x, w, b, sc)]
    y = F.conv1d(xr.transpose(1, 2), wr, br, padding=K // 2).transpose(1, 2) * m[:, :, None]
    if use_scale:
        y = F.relu(y * sr)
    gy = rnd(*y.shape, seed=4)
    y.backward(gy)
    xg, wg, bg, sg = g32(x, True), g32(w, True), g32(b, True), g32(sc, True)
    yo = ops.smallconv(xg, wg, bg, g32(m), sg if use_scale else None)
    close(yo, y, "smallconv fwd")
    yo.backward(g32(gy))
    close(xg.grad, xr.grad, "smallconv dx")
    close(wg.grad, wr.grad, "smallconv dw")
    close(bg.grad, br.grad, "smallconv dbias")
    if use_scale:
        close(sg.grad, sr.grad, "smallconv dscale")


def test_interp_concat_matches_reference_vectors(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "interp.npz"))
    ka = ops.interp_concat([torch.tensor(np.tile(g["known_in"][:, None], (1, 4))).to(dev())], 8)
    assert torch.equal(ka[:, 0].cpu(), torch.from_numpy(g["known_out"]))      # deepfake_video_audio.py:1726
    for i in range(int(g["n_cases"])):
        # channel counts in the fixture (6, 10, 4) are not multiples of 4: pad each stream to 12 channels
        streams, refs = [], []
        for key in ("v", "a", "e"):
            a = g[f"{key}{i}"]
            pad = np.zeros((a.shape[0], 12), np.float32)
            pad[:, :a.shape[1]] = a
            streams.append(torch.from_numpy(pad).to(dev()))
        out = ops.interp_concat(streams, 768).cpu().numpy()                     # [768, 36] time-major
        ref = g[f"out{i}"]                                                      # [20, 768] channel-major
        got = np.concatenate([out[:, 0:6], out[:, 12:22], out[:, 24:28]], axis=1).T
        assert np.array_equal(got, ref), f"interp case {i}: max diff {np.abs(got - ref).max():.3e}"


def test_adamw_ema_kernel(ops):
    from audio_visual_deepfake_detection_amd._lib import lib, check
    n = 10007
    p0, g0 = rnd(n, seed=1).float(), rnd(n, seed=2).float()
    pr = p0.clone().double().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    ema_ref = p0.clone().double()
    p, m, v, ema = p0.to(dev()), torch.zeros(n, device=dev()), torch.zeros(n, device=dev()), p0.to(dev())
    nsq = torch.zeros(1, device=dev())
    ws = torch.empty(int(lib.avdf_sumsq_ws()), device=dev())
    st = torch.cuda.current_stream().cuda_stream
    for step in range(1, 4):
        g = (g0 * step).double()
        gn = g.norm()
        clip = min(1.0 / (float(gn) + 1e-6), 1.0)
        pr.grad = g * clip
        opt.step()
        ema_ref = ema_ref * 0.999 + 0.001 * pr.detach()
        gd = (g0 * step).to(dev())
        nsq.zero_()
        check(lib.avdf_sumsq(gd.data_ptr(), nsq.data_ptr(), ws.data_ptr(), n, st), "sumsq")
        hp = torch.tensor([1e-3, 1 - 0.9 ** step, 1 - 0.999 ** step], device=dev())
        check(lib.avdf_adamw_ema(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), ema.data_ptr(),
                                 nsq.data_ptr(), hp.data_ptr(), n, 0.9, 0.999, 1e-8, 0.05, 1.0, 0.999, st), "adamw")
        close(nsq.sqrt(), gn.view(1), "grad norm")
        nsq2 = torch.zeros(1, device=dev())                     # a view that is not 16-byte aligned takes the scalar path
        check(lib.avdf_sumsq(gd[1:].data_ptr(), nsq2.data_ptr(), ws.data_ptr(), n - 1, st), "sumsq")
        close(nsq2.sqrt(), g[1:].norm().view(1), "grad norm (unaligned)")
    close(p, pr.detach(), "adamw params", rtol=1e-5)
    close(ema, ema_ref, "ema", rtol=1e-5)


# ------------------------------------------------------------------------------------------------
# fused LN -> depthwise k3 -> LN chains of q / k / v
@pytest.mark.parametrize("B,T,C,stride,share", [(2, 96, 256, 1, "all"), (3, 70, 64, 1, "kv"), (2, 64, 32, 2, "none"),
                                                (8, 768, 256, 1, "all"), (2, 130, 256, 2, "kv"), (1, 33, 8, 1, "all")])
@pytest.mark.parametrize("passthrough,main", [(False, False), (True, False), (True, True)])
def test_qkv_chain_matches_torch(ops, B, T, C, stride, share, passthrough, main):
    """passthrough: the op also hands x_q back (4th output) for the block's residual connection; its gradient must be
    summed into dx by the backward kernel (the reference graph simply uses x_q twice).  main: the engine path -- every
    parameter owns a `main_grad` buffer and the backward kernel ADDS its partial sums there (float atomics; branches that
    share a parameter meet in its buffer), autograd sees no parameter gradient."""
    if stride == 2 and T % 2:
        T += 1
    Tout = (T - 1) // stride + 1
    eps = 1e-5
    xs = [rnd(B, T, C, seed=1), rnd(B, T, C, seed=2), rnd(B, T, C, seed=3)]
    if share == "all":
        xs = [xs[0], xs[0], xs[0]]
    elif share == "kv":
        xs = [xs[0], xs[1], xs[1]]
    npar = 1 if share == "all" else 3
    la = [(1 + 0.2 * rnd(1, C, 1, seed=10 + i), 0.2 * rnd(1, C, 1, seed=20 + i)) for i in range(npar)]
    la = [la[0]] * 3 if npar == 1 else la
    dw = [rnd(C, 1, 3, seed=30 + i, scale=0.5) for i in range(3)]
    lb = [(1 + 0.2 * rnd(1, C, 1, seed=40 + i), 0.2 * rnd(1, C, 1, seed=50 + i)) for i in range(3)]
    mask = mk_mask(B, Tout, [Tout - 3 * b for b in range(B)])
    dys = [rnd(B, Tout, C, seed=60 + i) for i in range(3)]

    # float64 reference
    uniq = {}
    def leaf(t):
        if id(t) not in uniq:
            uniq[id(t)] = t.clone().requires_grad_(True)
        return uniq[id(t)]
    ref_y = []
    for i in range(3):
        x, (ga, ba), w, (gb, bb) = leaf(xs[i]), (leaf(la[i][0]), leaf(la[i][1])), leaf(dw[i]), (leaf(lb[i][0]), leaf(lb[i][1]))
        z = F.layer_norm(x, (C,), ga.view(-1), ba.view(-1), eps)
        u = F.conv1d(z.transpose(1, 2), w, None, stride, 1, groups=C).transpose(1, 2) * mask[..., None]
        ref_y.append(F.layer_norm(u, (C,), gb.view(-1), bb.view(-1), eps))
    dres = rnd(B, T, C, seed=70)
    (sum((y * d).sum() for y, d in zip(ref_y, dys)) + ((leaf(xs[0]) * dres).sum() if passthrough else 0.0)).backward()

    dmap = {}
    def dleaf(t):
        if id(t) not in dmap:
            dmap[id(t)] = g32(t, True)
        return dmap[id(t)]
    base = {}
    if main:
        for t in [a for pr in la for a in pr] + dw + [a for pr in lb for a in pr]:
            d = dleaf(t)
            if not hasattr(d, "main_grad"):
                d.main_grad = g32(rnd(*t.shape, seed=90 + len(base)))          # accumulate ON TOP of existing contents
                base[id(t)] = d.main_grad.clone()
    got = ops.qkv_chain([dleaf(x) for x in xs], [(dleaf(a), dleaf(b)) for a, b in la], [dleaf(w) for w in dw],
                        [(dleaf(a), dleaf(b)) for a, b in lb], g32(mask), stride, eps, passthrough=passthrough)
    assert len(got) == (4 if passthrough else 3)
    for i in range(3):
        close(got[i], ref_y[i], f"chain y[{i}]")
    loss = sum((y * g32(d)).sum() for y, d in zip(got[:3], dys))
    if passthrough:
        assert got[3].data_ptr() == dleaf(xs[0]).data_ptr()
        loss = loss + (got[3] * g32(dres)).sum()
    loss.backward()
    for key, t64 in uniq.items():
        if key in base:
            assert dmap[key].grad is None, "accumulate path must not hand a parameter gradient to autograd"
            close(dmap[key].main_grad - base[key], t64.grad, f"chain main_grad of tensor shape {tuple(t64.shape)}", rtol=5e-5)
        else:
            close(dmap[key].grad, t64.grad, f"chain grad of tensor shape {tuple(t64.shape)}", rtol=5e-5)


@pytest.mark.parametrize("B,T,C,bias,main", [(2, 96, 64, True, False), (8, 768, 256, True, True), (3, 50, 32, False, False)])
def test_conv1d_group3_matches_three_convs(ops, B, T, C, bias, main):
    xs = [rnd(B, T, C, seed=i) for i in range(3)]
    ws = [rnd(C, C, 1, seed=10 + i, scale=C ** -0.5) for i in range(3)]
    bs = [rnd(C, seed=20 + i) if bias else None for i in range(3)]
    dys = [rnd(B, T, C, seed=30 + i) for i in range(3)]
    x64 = [t.clone().requires_grad_(True) for t in xs]
    w64 = [t.clone().requires_grad_(True) for t in ws]
    b64 = [None if t is None else t.clone().requires_grad_(True) for t in bs]
    ref = [F.conv1d(x.transpose(1, 2), w, b).transpose(1, 2) for x, w, b in zip(x64, w64, b64)]
    sum((y * d).sum() for y, d in zip(ref, dys)).backward()
    xd, wd = [g32(t, True) for t in xs], [g32(t, True) for t in ws]
    bd = [None if t is None else g32(t, True) for t in bs]
    if main:          # engine path: kernels accumulate straight into `main_grad`
        for t in wd + [b for b in bd if b is not None]:
            t.main_grad = torch.zeros_like(t)
    got = ops.conv1d_group3(xd, wd, bd)
    for i in range(3):
        close(got[i], ref[i], f"group3 y[{i}]")
    sum((y * g32(d)).sum() for y, d in zip(got, dys)).backward()
    for i in range(3):
        close(xd[i].grad, x64[i].grad, f"group3 dx[{i}]")
        close(wd[i].main_grad if main else wd[i].grad, w64[i].grad, f"group3 dw[{i}]", rtol=5e-5)
        if bias:
            close(bd[i].main_grad if main else bd[i].grad, b64[i].grad, f"group3 db[{i}]", rtol=5e-5)


@pytest.mark.parametrize("B,T,C,masks,main", [(2, 96, 256, True, False), (8, 768, 256, True, True), (3, 50, 64, False, False),
                                             (2, 40, 512, True, False)])
def test_residual_ln_matches_torch(ops, B, T, C, masks, main):
    a, r = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2)
    scale, keep = rnd(1, C, 1, seed=3, scale=0.3), torch.tensor([1.0 / 0.9 if b % 3 else 0.0 for b in range(B)]).double()
    gamma, beta = 1 + 0.2 * rnd(1, C, 1, seed=4), 0.2 * rnd(1, C, 1, seed=5)
    ma = mk_mask(B, T, [T - 2 * b for b in range(B)]) if masks else None
    mr = mk_mask(B, T, [T - 3 * b for b in range(B)]) if masks else None
    dout, dh = rnd(B, T, C, seed=6), rnd(B, T, C, seed=7)
    leaves = [t.clone().requires_grad_(True) for t in (a, r, scale, gamma, beta)]
    a6, r6, s6, g6, b6 = leaves
    o6 = a6 * (ma[..., None] if masks else 1.0) + s6.view(1, 1, C) * keep.view(B, 1, 1) * (r6 * (mr[..., None] if masks else 1.0))
    h6 = F.layer_norm(o6, (C,), g6.view(-1), b6.view(-1), 1e-5)
    ((o6 * dout).sum() + (h6 * dh).sum()).backward()
    dl = [g32(t, True) for t in (a, r, scale, gamma, beta)]
    if main:
        for t in dl[2:]:
            t.main_grad = torch.zeros_like(t)
    out, h = ops.residual_ln(dl[0], None if ma is None else g32(ma), dl[1], None if mr is None else g32(mr), dl[2],
                             g32(keep), dl[3], dl[4], 1e-5)
    close(out, o6, "residual_ln out")
    close(h, h6, "residual_ln h")
    ((out * g32(dout)).sum() + (h * g32(dh)).sum()).backward()
    for i, name in enumerate(("da", "dr", "dscale", "dgamma", "dbeta")):
        got = dl[i].main_grad if (main and i >= 2) else dl[i].grad
        close(got, leaves[i].grad, f"residual_ln {name}", rtol=5e-5)


@pytest.mark.parametrize("B,T,C,H,masked", [(2, 96, 64, 256, True), (8, 768, 256, 1024, True), (3, 50, 32, 64, False)])
def test_mlp_residual_node_matches_torch(ops, B, T, C, H, masked):
    """The MLP half of a block as one autograd node on the engine path (ops.MlpResidualFn: fc1 -> GELU -> fc2 -> mask ->
    AffineDropPath -> + residual, blocks.py:1291-1297, 1313), the GELU derivative folded into the fc2 data-gradient
    epilogue (avdf_conv1d_bwd_pair_dact), against float64 autograd of the same expression."""
    x, a = rnd(B, T, C, seed=1), rnd(B, T, C, seed=2)
    w1, b1 = rnd(H, C, 1, seed=3, scale=C ** -0.5), rnd(H, seed=4)
    w2, b2 = rnd(C, H, 1, seed=5, scale=H ** -0.5), rnd(C, seed=6)
    scale = rnd(1, C, 1, seed=7, scale=0.3)
    keep = torch.tensor([1.0 / 0.9 if b % 3 else 0.0 for b in range(B)]).double()
    m = mk_mask(B, T, [T - 5 * b for b in range(B)]) if masked else None
    dout = rnd(B, T, C, seed=8)
    leaves = [t.clone().requires_grad_(True) for t in (x, a, w1, b1, w2, b2, scale)]
    x6, a6, w16, b16, w26, b26, s6 = leaves
    h6 = F.gelu(F.conv1d(x6.transpose(1, 2), w16, b16))
    y6 = F.conv1d(h6, w26, b26).transpose(1, 2)
    if masked:
        y6 = y6 * m[..., None]
    o6 = a6 + s6.view(1, 1, C) * keep.view(B, 1, 1) * y6
    (o6 * dout).sum().backward()
    dl = [g32(t, True) for t in (x, a, w1, b1, w2, b2, scale)]
    for t in dl[2:]:
        t.main_grad = torch.zeros_like(t)
    assert ops.mlp_residual_fusable(dl[0], dl[2], dl[3], dl[4], dl[5], dl[1])
    out = ops.mlp_residual(dl[0], dl[2], dl[3], dl[4], dl[5], g32(m) if masked else None, dl[1], dl[6], g32(keep))
    close(out, o6, "mlp node out")
    (out * g32(dout)).sum().backward()
    close(dl[0].grad, leaves[0].grad, "mlp node dx")
    close(dl[1].grad, leaves[1].grad, "mlp node da")
    for i, name in ((2, "dw1"), (3, "db1"), (4, "dw2"), (5, "db2"), (6, "dscale")):
        close(dl[i].main_grad, leaves[i].grad, f"mlp node {name}", rtol=5e-5)


@pytest.mark.parametrize("B,C,lens,main", [(2, 64, (32, 16, 8), True), (8, 256, (768, 384, 192, 96, 48, 24), True),
                                           (3, 32, (16, 8), False)])
def test_fpn_tail_matches_per_level_path(ops, B, C, lens, main):
    """FPN1D behind the lateral convs, all levels in two launches forward / three backward (ops.fpn_tail), against the
    per-level reference path in float64: top-down nearest x2 add, masked depthwise k3, LayerNorm (necks.py:62-93), laid out
    as the gapped level sequence of the shared heads.  main=False: forward only (inference)."""
    n = len(lens)
    lat = [rnd(B, T, C, seed=10 + l) for l, T in enumerate(lens)]
    msk = [mk_mask(B, T, [max(T - (b * 3) % T, 1) for b in range(B)]) for T in lens]
    dws = [rnd(C, 1, 3, seed=20 + l, scale=0.5) for l in range(n)]
    gam = [1 + 0.2 * rnd(1, C, 1, seed=30 + l) for l in range(n)]
    bet = [0.2 * rnd(1, C, 1, seed=40 + l) for l in range(n)]
    Tg = sum(lens) + n
    dy = rnd(B, Tg, C, seed=50)
    leaves = [[t.clone().requires_grad_(True) for t in grp] for grp in (lat, dws, gam, bet)]
    l6, w6, g6, b6 = leaves
    td = list(l6)
    for i in range(n - 1, 0, -1):
        td[i - 1] = td[i - 1] + td[i].repeat_interleave(2, dim=1)
    rows, mrows = [], []
    for l in range(n):
        u = F.conv1d(td[l].transpose(1, 2), w6[l], None, padding=1, groups=C).transpose(1, 2) * msk[l][..., None]
        rows += [F.layer_norm(u, (C,), g6[l].view(-1), b6[l].view(-1), 1e-5), torch.zeros(B, 1, C, dtype=torch.float64)]
        mrows += [msk[l], torch.zeros(B, 1, dtype=torch.float64)]
    ref, refm = torch.cat(rows, 1), torch.cat(mrows, 1)
    (ref * dy).sum().backward()
    dl = [[g32(t, main) for t in grp] for grp in (lat, dws, gam, bet)]
    if main:
        for grp in dl[1:]:
            for t in grp:
                t.main_grad = torch.zeros_like(t)
    import contextlib
    with (contextlib.nullcontext() if main else torch.no_grad()):
        assert ops.fpn_tail_fusable(dl[0], dl[1], dl[2], dl[3])
        y, gm = ops.fpn_tail(dl[0], [g32(m) for m in msk], dl[1], dl[2], dl[3], 1e-5)
    close(y, ref, "fpn tail y")
    close(gm, refm, "fpn tail mask")
    if not main:
        return
    (y * g32(dy)).sum().backward()
    for l in range(n):
        close(dl[0][l].grad, l6[l].grad, f"fpn tail dlat[{l}]")
        close(dl[1][l].main_grad, w6[l].grad, f"fpn tail ddw[{l}]", rtol=5e-5)
        close(dl[2][l].main_grad, g6[l].grad, f"fpn tail dgamma[{l}]", rtol=5e-5)
        close(dl[3][l].main_grad, b6[l].grad, f"fpn tail dbeta[{l}]", rtol=5e-5)


def test_interp_batched_equals_per_clip(ops):
    g = torch.Generator().manual_seed(4)
    bufs = [torch.randn(3, 50, 8, generator=g).cuda(), torch.randn(3, 23, 16, generator=g).cuda(), torch.randn(3, 96, 4, generator=g).cuda()]
    out = torch.empty(3, 96, 28, device="cuda")
    ops.interp_concat_batched(bufs, out)
    for b in range(3):
        assert torch.equal(out[b], ops.interp_concat([s[b] for s in bufs], 96))


@pytest.mark.parametrize("case", [(2, 96, 40, 32, 3, 1), (2, 72, 256, 256, 1, 1), (6, 768, 64, 256, 3, 1), (3, 96, 64, 36, 3, 2),
                                  (8, 768, 256, 1024, 1, 1), (1, 24, 2048, 256, 3, 2)])
def test_conv1d_forward_bf16x3_mode(ops, case):
    """Split-bf16 (hi*hi + hi*lo + lo*hi) forward GEMM of the inference paths: fp32-grade result (~5e-6 of the output
    scale) from the bf16 matrix cores; the default mode outside `ops.gemm_mode(1)` stays exact fp32."""
    B, T, Cin, Cout, K, s = case
    x, w, b = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, K, seed=2, scale=(Cin * K) ** -0.5), rnd(Cout, seed=3)
    ref = F.conv1d(x.transpose(1, 2), w, b, s, K // 2).transpose(1, 2)
    assert ops.current_gemm_mode() == 0
    with ops.gemm_mode(1):
        y1 = ops.conv1d(g32(x), g32(w), g32(b), None, s)
    assert ops.current_gemm_mode() == 0
    y0 = ops.conv1d(g32(x), g32(w), g32(b), None, s)
    close(y1, ref, "bf16x3 conv fwd", rtol=2e-5)
    close(y0, ref, "fp32 conv fwd", rtol=2e-6 if Cin * K < 4096 else 5e-6)
    # (plans whose K-group slice is not a multiple of 16 keep the exact kernel even in mode 1, so y0 may equal y1)


@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("shape", [(2, 200, 96, 136, 3, 1), (3, 192, 256, 64, 1, 1), (2, 96, 64, 40, 3, 2), (1, 520, 1024, 256, 1, 1)])
def test_conv1d_every_gemm_plan(ops, shape, mode):
    """Every (tile, K-groups, double-buffer) plan the tuner may select, forced through avdf_gemm_override, for the
    forward and the data-gradient kernels, against float64 -- in exact fp32 (mode 0) and with split-bf16 operands in
    both directions (mode 3: ~1e-5 of the output scale; plans whose K-group slice is not a multiple of 16 keep the exact
    kernel)."""
    from audio_visual_deepfake_detection_amd._lib import lib, check
    B, T, Cin, Cout, K, s = shape
    rt = RTOL if mode == 0 else 3e-5
    Tout = (T + 2 * (K // 2) - K) // s + 1
    x, w = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, K, seed=2, scale=(Cin * K) ** -0.5)
    dy = rnd(B, Tout, Cout, seed=3)
    x64, w64 = x.clone().requires_grad_(True), w.clone()
    ref = F.conv1d(x64.transpose(1, 2), w64, None, s, K // 2).transpose(1, 2)
    (ref * dy).sum().backward()
    xd, wd, dyd = g32(x), g32(w), g32(dy)
    st = torch.cuda.current_stream().cuda_stream
    try:
        ops.set_gemm_mode(mode)
        for tile in range(7):
            for kg in (1, 2, 4):
                if kg > (1, 2, 4, 4, 2, 4, 4)[tile]:
                    continue
                for db in (0, 1):
                    check(lib.avdf_gemm_override(0, tile, kg + 8 * db), "override")
                    check(lib.avdf_gemm_override(1, tile, kg + 8 * db), "override")
                    y = torch.empty(B, Tout, Cout, device=dev())
                    dx = torch.empty(B, T, Cin, device=dev())
                    check(lib.avdf_conv1d_fwd(xd.data_ptr(), wd.data_ptr(), None, None, y.data_ptr(), None, B, T, Cin, Tout, Cout,
                                              K, s, 0, st), "fwd")
                    check(lib.avdf_conv1d_bwd_data(dyd.data_ptr(), wd.data_ptr(), None, None, dx.data_ptr(), B, T, Cin, Tout, Cout,
                                                   K, s, 0, st), "dgrad")
                    close(y, ref, f"fwd tile {tile} kg {kg} db {db}", rtol=rt)
                    close(dx, x64.grad, f"dgrad tile {tile} kg {kg} db {db}", rtol=rt)
    finally:
        ops.set_gemm_mode(0)
        for k in range(3):
            check(lib.avdf_gemm_override(k, -1, -1), "override")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("n,B,T,Cin,Cout,act", [(1, 2, 200, 64, 136, 1), (3, 3, 192, 256, 256, 0), (1, 8, 96, 1024, 256, 0),
                                                (2, 1, 12, 256, 1024, 2)])
def test_conv1d_bwd_pair_every_plan(ops, n, B, T, Cin, Cout, act, mode):
    """avdf_conv1d_bwd_pair_grouped (K=1 dgrad + accumulating wgrad in one launch) for every dgrad plan the tuner may
    select, against the two separate entry points (dx bit-identical: same workgroup body) and against float64; mode 2 =
    the split-bf16 backward kernels."""
    import ctypes
    from audio_visual_deepfake_detection_amd._lib import lib, check
    tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    st = torch.cuda.current_stream().cuda_stream
    xs = [g32(rnd(B, T, Cin, seed=10 + i)) for i in range(n)]
    ws_ = [g32(rnd(Cout, Cin, 1, seed=20 + i, scale=Cin ** -0.5)) for i in range(n)]
    dys = [g32(rnd(B, T, Cout, seed=30 + i)) for i in range(n)]
    ypres = [g32(rnd(B, T, Cout, seed=40 + i)) for i in range(n)]
    mask = (torch.arange(T)[None, :] < torch.tensor([T - 3 * b for b in range(B)])[:, None]).float().to(dev()).contiguous()
    masks = [mask] * n
    # float64 reference: g = dy * mask * act'(ypre); dx = g @ W; dw = g^T x; db = colsum(g)
    def gref(i):
        g = dys[i].double() * mask.double()[..., None]
        p = ypres[i].double()
        if act == 1:
            g = g * (p > 0)
        elif act == 2:
            g = g * (0.5 * (1 + torch.erf(p / 2 ** 0.5)) + p * torch.exp(-0.5 * p * p) / (2 * torch.pi) ** 0.5)
        return g
    base_dw = [g32(rnd(Cout, Cin, 1, seed=50 + i)) for i in range(n)]
    base_db = [g32(rnd(Cout, seed=60 + i)) for i in range(n)]
    # separate launches (the reference point)
    dx_sep = [torch.empty(B, T, Cin, device=dev()) for _ in range(n)]
    rt = RTOL if mode == 0 else 3e-5
    try:
        ops.set_gemm_mode(mode)
        check(lib.avdf_gemm_override(4, -2, -2), "override")      # the pair's own table rows off: plans come from kind 1 / 2
        for tile in range(7):
            for kg in (1, 2, 4):
                if kg > (1, 2, 4, 4, 2, 4, 4)[tile] or (tile == 6 and kg == 1):     # 64x32 tile: the fused launch needs kg >= 2
                    continue
                check(lib.avdf_gemm_override(1, tile, kg), "override")
                check(lib.avdf_conv1d_bwd_data_grouped(n, tab(dys), tab(ws_), tab(masks), tab(ypres), tab(dx_sep), B, T, Cin, T, Cout,
                                                       1, 1, act, st), "dgrad")
                dx = [torch.full((B, T, Cin), float("nan"), device=dev()) for _ in range(n)]
                dw = [b.clone() for b in base_dw]
                db = [b.clone() for b in base_db]
                check(lib.avdf_conv1d_bwd_pair_grouped(n, tab(dys), tab(ws_), tab(xs), tab(masks), tab(ypres), tab(dx), tab(dw), tab(db),
                                                       B, T, Cin, Cout, act, st), "pair")
                for i in range(n):
                    g = gref(i)
                    assert torch.equal(dx[i], dx_sep[i]), f"dx differs from the separate dgrad launch (tile {tile} kg {kg})"
                    close(dx[i], (g @ ws_[i][:, :, 0].double()).cpu(), f"dx tile {tile} kg {kg}", rtol=rt)
                    close(dw[i][:, :, 0] - base_dw[i][:, :, 0],
                          torch.einsum("btn,btc->nc", g, xs[i].double()).cpu(), f"dw tile {tile} kg {kg}", rtol=2e-4)
                    close(db[i] - base_db[i], g.sum((0, 1)).cpu(), f"db tile {tile} kg {kg}", rtol=2e-4)
        # fusion off: same entry, two launches
        check(lib.avdf_gemm_override(1, -1, -1), "override")
        check(lib.avdf_bwd_pair_mode(0), "mode")
        dx = [torch.empty(B, T, Cin, device=dev()) for _ in range(n)]
        dw = [torch.zeros(Cout, Cin, 1, device=dev()) for _ in range(n)]
        check(lib.avdf_conv1d_bwd_pair_grouped(n, tab(dys), tab(ws_), tab(xs), tab(masks), tab(ypres), tab(dx), tab(dw), None,
                                               B, T, Cin, Cout, act, st), "pair off")
        for i in range(n):
            close(dw[i][:, :, 0], torch.einsum("btn,btc->nc", gref(i), xs[i].double()).cpu(), "dw (pair off)", rtol=2e-4)
    finally:
        ops.set_gemm_mode(0)
        check(lib.avdf_bwd_pair_mode(1), "mode")
        for k in (0, 1, 2, 4):
            check(lib.avdf_gemm_override(k, -1, -1), "override")


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,Cin,Cout,use_keep,masked", [(2, 96, 1024, 256, True, True), (8, 768, 64, 32, False, True),
                                                         (3, 12, 256, 256, True, False), (1, 50, 36, 20, True, True)])
def test_conv1d_residual_matches_separate_ops(ops, B, T, Cin, Cout, use_keep, masked):
    """fc2 + AffineDropPath + residual in one launch (ops.conv1d_residual) == conv1d followed by affine_residual, and
    == float64, forward and every gradient."""
    x, w, bias = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, 1, seed=2, scale=Cin ** -0.5), rnd(Cout, seed=3, scale=0.1)
    a, scale = rnd(B, T, Cout, seed=4), rnd(1, Cout, 1, seed=5, scale=0.5)
    keep = (torch.tensor([1.25, 0.0, 1.25, 1.25, 0.0, 1.25, 1.25, 1.25][:B], dtype=torch.float64) if use_keep else None)
    mask = mk_mask(B, T, [T - 2 * b for b in range(B)]) if masked else None
    dout = rnd(B, T, Cout, seed=6)
    # float64 reference
    leaves = [t.clone().requires_grad_(True) for t in (x, w, bias, a, scale)]
    x6, w6, b6, a6, s6 = leaves
    h = F.conv1d(x6.transpose(1, 2), w6, b6).transpose(1, 2)
    if mask is not None:
        h = h * mask[..., None]
    ref = a6 + s6.view(1, 1, -1) * (keep.view(-1, 1, 1) if keep is not None else 1.0) * h
    (ref * dout).sum().backward()
    dl = [g32(t, True) for t in (x, w, bias, a, scale)]
    got = ops.conv1d_residual(dl[0], dl[1], dl[2], None if mask is None else g32(mask), dl[3], dl[4],
                              None if keep is None else g32(keep))
    close(got, ref, "conv1d_residual out")
    (got * g32(dout)).sum().backward()
    for t32, t64, name in zip(dl, leaves, ("x", "w", "bias", "a", "scale")):
        close(t32.grad, t64.grad, f"conv1d_residual d{name}", rtol=5e-5)
    # the separate ops give the same forward values (same GEMM kernel, same epilogue arithmetic order)
    with torch.no_grad():
        sep = ops.affine_residual(dl[3], None, ops.conv1d(dl[0], dl[1], dl[2], None if mask is None else g32(mask)), None, dl[4],
                                  None if keep is None else g32(keep))
        fused = ops.conv1d_residual(dl[0], dl[1], dl[2], None if mask is None else g32(mask), dl[3], dl[4],
                                    None if keep is None else g32(keep))
    close(fused, sep.double().cpu(), "fused vs separate", rtol=1e-6)


# ------------------------------------------------------------------------------------------------
# pyramid glue (csrc/glue.hip): gapped level sequence, head-output compaction with Scale + ReLU, mask pyramid
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,lens", [(2, 32, [48, 24, 12]), (3, 256, [768, 384, 192, 96, 48, 24]), (1, 8, [5])])
def test_gap_concat_matches_cat_with_zero_rows(ops, B, C, lens):
    xs = [rnd(B, n, C, seed=10 + i) for i, n in enumerate(lens)]
    ms = [mk_mask(B, n, [n - (b % 3) for b in range(B)]) for n in lens]
    xg = [g32(x, True) for x in xs]
    out, gmask = ops.gap_concat(xg, [g32(m) for m in ms])
    z, zm = torch.zeros(B, 1, C, dtype=torch.float64), torch.zeros(B, 1, dtype=torch.float64)
    ref = torch.cat([t for x in xs for t in (x, z)], dim=1)
    refm = torch.cat([t for m in ms for t in (m, zm)], dim=1)
    close(out, ref.float(), "gap_concat", rtol=0)               # pure data movement: bit-exact on the fp32 values
    close(gmask, refm, "gap mask", rtol=0)
    w = rnd(*ref.shape, seed=99)
    (out * g32(w)).sum().backward()
    off = 0
    for x, n in zip(xg, lens):
        close(x.grad, w[:, off:off + n].float(), "gap_concat bwd", rtol=0)
        off += n + 1


@pytest.mark.parametrize("B,NC,lens,relu", [(2, 2, [48, 24, 12], True), (8, 2, [768, 384, 192, 96, 48, 24], True),
                                            (3, 1, [96, 48], False)])
def test_head_out_matches_index_select_scale_relu(ops, B, NC, lens, relu):
    Tg = sum(lens) + len(lens)
    y = rnd(B, Tg, NC, seed=3)
    sc = [1.0 + 0.2 * float(rnd(1, seed=40 + l)) for l in range(len(lens))] if relu else None
    idx, rs, off = [], [], 0
    for l, n in enumerate(lens):
        idx.append(torch.arange(off, off + n))
        rs += [sc[l] if sc else 1.0] * (n + 1)
        off += n + 1
    idx = torch.cat(idx)
    yr = y.clone().requires_grad_(True)
    scr = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in sc] if sc else []
    if sc:
        rvec = torch.cat([s.reshape(1).expand(n + 1) for s, n in zip(scr, lens)])
        ref = torch.relu(yr * rvec[None, :, None]).index_select(1, idx)
    else:
        ref = yr.index_select(1, idx)
    yg = g32(y, True)
    scg = [torch.tensor(v, dtype=torch.float32, device=dev(), requires_grad=True) for v in sc] if sc else []
    out = ops.head_out(yg, lens, relu=relu, scales=scg)
    close(out, ref, "head_out")
    w = rnd(*ref.shape, seed=5)
    (ref * w).sum().backward()
    (out * g32(w)).sum().backward()
    close(yg.grad, yr.grad, "head_out dy")
    for a, b in zip(scg, scr):
        close(a.grad.reshape(1), b.grad.reshape(1), "head_out dscale", rtol=1e-4)


def test_mask_pyramid_equals_strided_slices(ops):
    m = mk_mask(4, 768, [768, 700, 1, 333])
    outs = ops.mask_pyramid(g32(m), 5, 2)
    cur = m
    for o in outs:
        cur = cur[:, ::2]
        close(o, cur, "mask pyramid", rtol=0)


def test_keep_factors_from_uniform_draws(ops):
    u = torch.rand(36, 8)
    kp = torch.tensor([0.9] * 30 + [0.5] * 6)
    got = ops.droppath_keep(u.to(dev()), kp.to(dev())[:, None])
    ref = (kp[:, None] + u).floor() / kp[:, None]
    assert torch.equal(got.cpu(), ref)
    u2 = torch.rand(8, 256)
    assert torch.equal(ops.dropout_keep(u2.to(dev()), 0.5).cpu(), (u2 >= 0.5).float() / 0.5)


def test_local_attention_rel_pe_and_time_weight_match_reference(golden_dir):
    """LocalMaskedMHCA with use_rel_pe AND use_time_weight (blocks.py:883-1224): output, input gradient and the gradients of
    both per-head tables against the reference class's own result (tests/golden/local_attention_tables.npz)."""
    from audio_visual_deepfake_detection_amd.libs.modeling.blocks import LocalMaskedMHCA
    g = np.load(os.path.join(golden_dir, "local_attention_tables.npz"))
    B, C, T, H, W = [int(v) for v in g["dims"]]
    rs = np.random.RandomState(int(g["seed"]))
    m = LocalMaskedMHCA(C, H, window_size=W, n_qx_stride=1, n_kv_stride=1, use_rel_pe=True, use_time_weight=True)
    sd = {}
    for k, v in m.state_dict().items():          # same key order and draws as the generator (reference state_dict order)
        if k.endswith("norm.weight") or k == "time_weighting":
            sd[k] = torch.from_numpy((1.0 + 0.2 * rs.standard_normal(v.shape)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((0.3 * rs.standard_normal(v.shape)).astype(np.float32))
    m.load_state_dict(sd)
    m = m.to(dev())
    x = torch.from_numpy(rs.standard_normal((B, C, T)).astype(np.float32))
    w = torch.from_numpy(rs.standard_normal((B, C, T)).astype(np.float32))
    mask = torch.ones(B, T)
    mask[1, 30:] = 0
    xt = x.transpose(1, 2).contiguous().to(dev()).requires_grad_(True)
    out, _ = m(xt, mask.to(dev()))
    (out * w.transpose(1, 2).contiguous().to(dev())).sum().backward()
    close(out.transpose(1, 2), torch.from_numpy(g["out"]), "local attention (rel_pe + time weight)")
    close(xt.grad.transpose(1, 2), torch.from_numpy(g["dx"]), "dx")
    close(m.rel_pe.grad, torch.from_numpy(g["drel"]), "d rel_pe", rtol=1e-4)
    close(m.time_weighting.grad, torch.from_numpy(g["dtw"]), "d time_weighting", rtol=1e-4)
    close(m.query.weight.grad, torch.from_numpy(g["dqw"]), "d query.weight", rtol=1e-4)


def test_tuned_extra_rows_swap_plans_at_run_time(ops):
    """avdf_tuned_extra (tools/plan_search.py): plan rows set at run time are honoured for exactly their shape -- every
    forward tile / K-group plan gives the float64 result -- and clearing them restores the table."""
    import ctypes
    from audio_visual_deepfake_detection_amd._lib import lib, check
    B, T, Cin, Cout = 3, 64, 256, 256
    x, w = rnd(B, T, Cin, seed=1), rnd(Cout, Cin, 1, seed=2, scale=Cin ** -0.5)
    ref = F.conv1d(x.transpose(1, 2), w).transpose(1, 2)
    xd, wd = g32(x), g32(w)
    try:
        for tile, kg in [(0, 1), (1, 2), (2, 4), (3, 4), (5, 2), (6, 4)]:
            row = f"0,1,1,{B * T},{Cout},{Cin},{tile},{kg}".encode()
            check(lib.avdf_tuned_extra(ctypes.c_char_p(row)), "tuned_extra")
            close(ops.conv1d(xd, wd), ref, f"forward with the run-time row ({tile},{kg})")
        check(lib.avdf_tuned_extra(ctypes.c_char_p(b"not a row")), "tuned_extra")       # unparsable text = no rows
        close(ops.conv1d(xd, wd), ref, "forward after a rejected row string")
    finally:
        check(lib.avdf_tuned_extra(None), "tuned_extra")
    close(ops.conv1d(xd, wd), ref, "forward with the table restored")


@pytest.mark.parametrize("shift,stride", [(1, 1), (3, 1), (-1, 1), (-2, 1), (2, 2)])
def test_qkv_chain_input_row_map_equals_explicit_resampling(ops, shift, stride):
    """qkv_chain(xshift=(0, s, s)): the k / v chains read their input through a nearest-resampling row map (the
    up / down-sampling of the cross-level blocks, backbones.py:487-490) -- forward outputs and every gradient must equal
    the explicit ops.upsample / ops.downsample followed by the plain chain."""
    B, T, C = 2, 96, 64
    Tsrc = (T >> shift) if shift > 0 else (T << -shift)
    Tout = (T - 1) // stride + 1
    xq, xkv = rnd(B, T, C, seed=1), rnd(B, Tsrc, C, seed=2)
    la = [(1 + 0.2 * rnd(1, C, 1, seed=10 + i), 0.2 * rnd(1, C, 1, seed=20 + i)) for i in range(3)]
    dw = [rnd(C, 1, 3, seed=30 + i, scale=0.5) for i in range(3)]
    lb = [(1 + 0.2 * rnd(1, C, 1, seed=40 + i), 0.2 * rnd(1, C, 1, seed=50 + i)) for i in range(3)]
    mask = g32(mk_mask(B, Tout, [Tout, Tout - 5]))
    dys = [g32(rnd(B, Tout, C, seed=60 + i)) for i in range(3)]

    def run(mapped):
        leaves = [g32(xq, True), g32(xkv, True)] + [g32(t, True) for pr in la for t in pr] + [g32(t, True) for t in dw] + \
                 [g32(t, True) for pr in lb for t in pr]
        q, kv = leaves[0], leaves[1]
        lna = [(leaves[2 + 2 * i], leaves[3 + 2 * i]) for i in range(3)]
        dws = leaves[8:11]
        lnb = [(leaves[11 + 2 * i], leaves[12 + 2 * i]) for i in range(3)]
        if mapped:
            ys = ops.qkv_chain([q, kv, kv], lna, dws, lnb, mask, stride, 1e-5, xshift=(0, shift, shift))
        else:
            r = 1 << abs(shift)
            kvr = ops.upsample(kv, r) if shift > 0 else ops.downsample(kv, r)
            ys = ops.qkv_chain([q, kvr, kvr], lna, dws, lnb, mask, stride, 1e-5)
        sum((y * d).sum() for y, d in zip(ys[:3], dys)).backward()
        return [y.detach() for y in ys[:3]], [t.grad for t in leaves]

    y1, g1 = run(True)
    y0, g0 = run(False)
    for a, b in zip(y1, y0):
        assert torch.equal(a, b), "forward differs from the explicit resampling"
    for i, (a, b) in enumerate(zip(g1, g0)):
        close(a, b.cpu(), f"gradient of leaf {i}", rtol=1e-6)


@pytest.mark.parametrize("scenario", ["trained_like", "dense_random", "few", "none"])
def test_decode_nms_voting_dense_candidates_match_oracle(ops, scenario):
    """Decode + hard NMS + voting on the device (one workgroup per clip, csrc/decode_nms.hip) against the oracle's
    restatement of inference_single_video + batched_nms (av_fd_no_recon.py:760-825, libs/utils/nms.py:103-190) with
    1512 points per clip: a trained detector's clustered candidates (long runs of suppressed ranks between survivors),
    every point a candidate, a handful, and none.  Kept points identical, segments to fp32 rounding."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import ummaformer_ref as R
    B, lens = 3, (768, 384, 192, 96, 48, 24)
    P = sum(lens)
    g = torch.Generator().manual_seed(11)
    pts_levels = []
    for l, T in enumerate(lens):
        s = float(2 ** l)
        pts_levels.append(torch.stack([torch.arange(T, dtype=torch.float32) * s, torch.zeros(T), torch.full((T,), 1e4),
                                       torch.full((T,), s)], 1))
    points = torch.cat(pts_levels)
    offsets = torch.rand(B, P, 2, generator=g) * 4 + 0.5
    if scenario == "trained_like":
        logits = torch.full((B, P), -7.5)
        for b in range(B):
            for k in range(3 + b):
                c = float(torch.randint(50, 700, (1,), generator=g))
                w = float(torch.randint(10, 80, (1,), generator=g))
                near = (points[:, 0] - c).abs() < w
                logits[b, near] = torch.randn(int(near.sum()), generator=g) + 2.0
                offsets[b, near, 0] = ((points[near, 0] - (c - w)) / points[near, 3]).clamp(min=0.1)
                offsets[b, near, 1] = (((c + w) - points[near, 0]) / points[near, 3]).clamp(min=0.1)
    elif scenario == "dense_random":
        logits = torch.randn(B, P, generator=g)
        logits[1, 100:140] = logits[1, 100]                       # ties: rank by point index
    elif scenario == "few":
        logits = torch.full((B, P), -9.0)
        logits[:, [5, 6, 7, 800, 1400]] = torch.tensor([1.0, 0.5, 0.2, 3.0, 0.0])
    else:
        logits = torch.full((B, P), -9.0)
    mask = torch.ones(B, P)
    mask[2, 700:768] = 0
    cfg = dict(num_classes=1, pre_nms_thresh=0.001, pre_nms_topk=5000, duration_thresh=0.05, nms_method="hard",
               multiclass_nms=False, min_score=0.001, iou_threshold=0.75, max_seg_num=100, voting_thresh=0.75)
    segs, scores, pidx, count = ops.decode_nms(points.to(dev()), logits.to(dev()), offsets.to(dev()), mask.to(dev()),
                                               cfg["pre_nms_thresh"], cfg["duration_thresh"], cfg["min_score"],
                                               cfg["iou_threshold"], cfg["voting_thresh"], cfg["max_seg_num"])
    torch.cuda.synchronize()
    for b in range(B):
        lg, of, mk, o = [], [], [], 0
        for T in lens:
            lg.append(logits[b, o:o + T, None]); of.append(offsets[b, o:o + T]); mk.append(mask[b, o:o + T]); o += T
        s_, c_, l_, p_ = R.decode_single(cfg, pts_levels, mk, lg, of)
        rs, rc, _, rk = R.batched_nms_ref(cfg, s_, c_, l_)
        n = int(count[b])
        assert n == rs.shape[0], f"{scenario} clip {b}: {n} segments kept, oracle {rs.shape[0]}"
        if n == 0:
            continue
        assert torch.equal(pidx[b, :n].cpu().long(), p_[rk]), f"{scenario} clip {b}: kept points differ"
        close(scores[b, :n], rc, f"{scenario} clip {b}: scores", rtol=1e-6)      # expf vs torch.sigmoid: an ulp
        close(segs[b, :n], rs, f"{scenario} clip {b}: voted segments", rtol=5e-6)


def test_reduce_segments_batched_matches_torch(ops):
    """avdf_reduce_segments_batched (the engine's one-launch fold of the parameter partials of every chain backward of a
    backward phase): 13 jobs (two launches) of different row counts, segments with one to three sources, ADDED to the
    destinations, against float64."""
    import ctypes
    from audio_visual_deepfake_detection_amd._lib import lib, check
    g = torch.Generator().manual_seed(5)
    C = 64
    jobs, refs, keep = [], [], []
    for j in range(13):
        nrow, stride = int(torch.randint(1, 700, (1,), generator=g)), 21 * C
        ws = torch.randn(nrow, stride, generator=g)
        segs = [(C, [0]), (C, [C, 8 * C]), (3 * C, [2 * C]), (C, [5 * C, 12 * C, 19 * C]), (4, [20 * C])][: 2 + j % 4]
        dst0 = [torch.randn(n, generator=g) for n, _ in segs]
        ref = [d.double() + sum(ws[:, o:o + n].double().sum(0) for o in src) for d, (n, src) in zip(dst0, segs)]
        wsd, dstd = g32(ws), [g32(d) for d in dst0]
        keep.append((wsd, dstd))
        jobs.append((wsd, nrow, stride, dstd, [n for n, _ in segs], [src for _, src in segs]))
        refs.append(ref)
    ci = lambda v: (ctypes.c_int * len(v))(*v)
    tab = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    srcs = [s_ for j in jobs for s_ in j[5]]
    check(lib.avdf_reduce_segments_batched(len(jobs), tab([j[0] for j in jobs]), ci([j[1] for j in jobs]), ci([j[2] for j in jobs]),
                                           ci([len(j[3]) for j in jobs]), tab([d for j in jobs for d in j[3]]),
                                           ci([n for j in jobs for n in j[4]]), ci([len(s_) for s_ in srcs]),
                                           ci([(s_ + [0, 0, 0])[k] for s_ in srcs for k in range(3)]),
                                           torch.cuda.current_stream().cuda_stream), "reduce_segments_batched")
    torch.cuda.synchronize()
    for j, (job, ref) in enumerate(zip(jobs, refs)):
        for s_, (d, r) in enumerate(zip(job[3], ref)):
            close(d, r, f"job {j} segment {s_}", rtol=1e-5)


def test_gelu_and_its_derivative_pinned_to_float64_erf(ops):
    """common.h's gelu_f / gelu_grad_f (Abramowitz-Stegun 7.1.26 CDF on the hardware exp / rcp; every GELU epilogue and
    derivative of the library, training and inference) against the exact-erf GELU of the reference (nn.GELU,
    libs/modeling/blocks.py:1291) in float64 over [-12, 12].  Stated bounds: 8e-7 ABSOLUTE on gelu(x) (measured 6.2e-7 on MI355X), 1.5e-6 absolute on
    gelu'(x) -- the tail x < -3 is NOT held to a relative bound (the rational CDF's relative error there is 1e-3..1e-2 of
    values below 4e-3).  The identity-weight K = 1 conv makes y = gelu(x) and dx = dy * gelu'(x) exactly."""
    C = 64
    xs = torch.linspace(-12.0, 12.0, 4096 * C, dtype=torch.float64)
    xs = torch.cat([xs, torch.tensor([0.0, -0.0, 1e-30, -1e-30] * (C // 4), dtype=torch.float64)]).view(1, -1, C)
    w = torch.eye(C, dtype=torch.float64).view(C, C, 1)
    xg = g32(xs, True)
    y = ops.conv1d(xg, g32(w), None, None, 1, ops.ACT_GELU)
    y.backward(torch.ones_like(y))
    ref = 0.5 * xs * (1.0 + torch.erf(xs / math.sqrt(2.0)))
    dref = 0.5 * (1.0 + torch.erf(xs / math.sqrt(2.0))) + xs * torch.exp(-0.5 * xs * xs) / math.sqrt(2.0 * math.pi)
    e = float((y.detach().cpu().double() - ref).abs().max())
    de = float((xg.grad.cpu().double() - dref).abs().max())
    print(f"gelu_f max abs error {e:.3e}; gelu_grad_f {de:.3e}")
    assert e <= 8e-7, f"gelu_f: max abs error {e:.3e}"
    assert de <= 1.5e-6, f"gelu_grad_f: max abs error {de:.3e}"
