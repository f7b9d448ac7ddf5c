"""PyTorch-ROCm custom ops over the C ABI of libavdf_hip.so.

PyTorch is plumbing here: it owns device memory (caching allocator), the HIP stream and the
autograd tape.  Every op below launches hand-written gfx950 kernels through ctypes; there is no
eager/aten fallback -- a CPU tensor or a missing library raises.

Layout: activations time-major [B, T, C] fp32; masks fp32 [B, T]; weights in the reference
layouts (Conv1d [Cout, Cin/groups, K], LayerNorm [1, C, 1], ...).
"""
import ctypes
import os as _os

import torch

from ._lib import lib, check

ACT_NONE, ACT_RELU, ACT_GELU, ACT_LEAKY = 0, 1, 2, 3
WGRAD_SIDE_STREAM = False  # engine.TrainStep: weight-gradient GEMMs run on a second HIP stream, overlapping the dgrad chain
FUSE_QKV_CHAIN = True      # LN -> depthwise conv -> LN of q, k, v as one fused launch (blocks._MHCABase.attend_prenorm)

# fork/join of independent branches on side streams (blocks._MHCABase._attend); off by default
BRANCH_STREAMS = False
_side = {}


def side_streams(device):
    key = str(device)
    if key not in _side:
        _side[key] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
    return _side[key]


# optional per-launch timing of the MFMA conv kernels (bench.py roofline pass): list of
# (kernel tag, algorithmic FLOPs, start event, end event) recorded on the launch stream
_prof = None


def profile_begin():
    global _prof
    _prof = []


def profile_end():
    global _prof
    out, _prof = _prof, None
    return out


shape_log = None     # tools/autotune.py: dict (kind, K, stride, B, Tin, Cin, Cout, nprob) -> launches per step


def _timed(tag, flops, fn, shape=None):
    if shape_log is not None and shape is not None:
        shape_log[shape] = shape_log.get(shape, 0) + 1
    if _prof is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    _prof.append((tag, flops, s, e, shape))
    return r


def _conv_tag(kind, K, stride, B, T, N):
    """Kernel class of a profile record: direction + kernel size + stride (the tile / K-group variant inside a class is
    a plan choice -- table or heuristic -- and is not part of the label)."""
    return f"conv_{kind}_k{K}s{stride}"


def _stream():
    return torch.cuda.current_stream().cuda_stream


_wg_side = {}
_wg_used = [False]


class _WgradSide:
    """Context for a weight-gradient launch of the accumulate path: its result (atomics into the flat gradient buffer)
    is only read by the optimiser, so it does not have to sit on the critical dgrad chain.  The side stream waits for
    the producer of `dy` and is joined ONCE, at the end of the backward pass (`wgrad_join`)."""

    def __init__(self, tensors):
        self.on = WGRAD_SIDE_STREAM and tensors[0].is_cuda
        self.tensors = tensors

    def __enter__(self):
        if not self.on:
            return self
        dev = self.tensors[0].device
        key = (dev.type, dev.index)
        if key not in _wg_side:
            _wg_side[key] = torch.cuda.Stream(device=dev)
        self.side = _wg_side[key]
        self.side.wait_stream(torch.cuda.current_stream())
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        _wg_used[0] = True
        return self

    def __exit__(self, *a):
        if not self.on:
            return False
        for t in self.tensors:
            if t is not None:
                t.record_stream(self.side)
        self.ctx.__exit__(*a)
        return False


def wgrad_join():
    """Make the current stream wait for every side-stream weight-gradient launch issued so far."""
    if _wg_used[0]:
        cur = torch.cuda.current_stream()
        for st in _wg_side.values():
            cur.wait_stream(st)
        _wg_used[0] = False


def _p(t):
    """Device pointer of a contiguous fp32 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("avdf ops need HIP (cuda) tensors: there is no CPU fallback")
    if t.dtype != torch.float32 and t.dtype != torch.int32:
        raise RuntimeError(f"avdf ops need float32/int32 tensors, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("avdf ops need contiguous tensors")
    return t.data_ptr()


def _c(t):
    return None if t is None else t.contiguous()


def _ptr_table(ts):
    """Host array of device pointers (the grouped C-ABI entries take `const float* const*` tables)."""
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _ws(n, like):
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=like.device)


# Accumulating segmented reductions (parameter partials of the chain backward kernels) queued while a backward phase of
# engine.TrainStep runs, then folded in ceil(n / 10) launches instead of one dependent tiny launch per chain.
_DEFERRED = None
DEFER_REDUCTIONS = _os.environ.get("AVDF_DEFER_REDUCTIONS", "1") != "0"
# ... and the row kernels (LayerNorm / AffineDropPath backward) hand their workgroups' partial rows to the same fold instead of
# adding them by float atomics inside the kernel
DEFER_ROW_PARTIALS = _os.environ.get("AVDF_DEFER_ROWS", "1") != "0"


def begin_deferred_reductions():
    global _DEFERRED
    if DEFER_REDUCTIONS:
        _DEFERRED = []


def abort_deferred_reductions():
    """The backward phase raised: drop the queued jobs (their destinations hold a partial gradient either way) so that
    the error that surfaces is the phase's own, not a fold over half-written workspaces."""
    global _DEFERRED
    _DEFERRED = None


def _deferrable(stride, lens, srcs=()):
    """Constraints of avdf_reduce_segments_batched (chain.hip), checked where a job is QUEUED so that an unsupported width
    takes the immediate reduction of its call site instead of failing after the phase's kernels have run: row stride
    <= 65535 floats, segment lengths / source offsets multiples of 4, <= 16 segments, <= 65535 columns per job."""
    return (_DEFERRED is not None and 0 < stride <= 65535 and stride % 4 == 0 and 0 < len(lens) <= 16
            and all(n > 0 and n % 4 == 0 for n in lens) and sum(lens) <= 65535
            and all(o % 4 == 0 for s_ in srcs for o in s_))


def flush_deferred_reductions():
    """Launch the queued jobs on the current stream (the stream the backward pass ran on) and stop queueing."""
    global _DEFERRED
    jobs, _DEFERRED = _DEFERRED, None
    if not jobs:
        return
    n = len(jobs)
    dst = [d for j in jobs for d in j[3]]
    lens = [v for j in jobs for v in j[4]]
    srcs = [s_ for j in jobs for s_ in j[5]]
    ci = lambda vals: (ctypes.c_int * len(vals))(*vals)
    check(lib.avdf_reduce_segments_batched(n, _ptr_table([j[0] for j in jobs]), ci([j[1] for j in jobs]), ci([j[2] for j in jobs]),
                                           ci([len(j[3]) for j in jobs]), _ptr_table(dst), ci(lens), ci([len(s_) for s_ in srcs]),
                                           ci([(s_ + [0, 0, 0])[k] for s_ in srcs for k in range(3)]), _stream()),
          "reduce_segments_batched")


def _affine_residual_bwd(dout, ma, r, mr, sc, keep, da, dr, ds, B, T, C, acc, like):
    """avdf_affine_residual_bwd; on the engine path the workgroups' dscale rows join the deferred fold of the backward phase
    instead of float atomics inside the kernel."""
    ws = _ws(lib.avdf_affine_residual_bwd_ws(B, T, C), like) if ds is not None else None
    rows = int(lib.avdf_affine_residual_bwd_rows(B, T, C)) if (acc and ds is not None and DEFER_ROW_PARTIALS
                                                               and _deferrable(C, [C])) else 0
    check(lib.avdf_affine_residual_bwd(_p(dout), _p(ma), _p(r), _p(mr), _p(sc), _p(keep), _p(da), _p(dr), _p(ds), _p(ws), B, T,
                                       C, 2 if rows else acc, _stream()), "affine_residual_bwd")
    if rows:
        _DEFERRED.append((ws, rows, C, [ds], [C], [[0]]))


def _main_grad(t):
    """Flat-buffer gradient view attached by engine.TrainStep (`param.main_grad`): when present, the
    backward kernels ACCUMULATE straight into it (accumulate=1) and autograd receives None, so there
    is no temporary gradient tensor and no aten add per parameter."""
    return None if t is None else getattr(t, "main_grad", None)


# --------------------------------------------------------------------------------------------
# dense conv1d (implicit GEMM, exact-fp32 MFMA)
# --------------------------------------------------------------------------------------------
class Conv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, mask, stride, act):
        w0, b0 = w, bias
        x, w, bias, mask = _c(x), _c(w), _c(bias), _c(mask)
        B, Tin, Cin = x.shape
        Cout, Cin_w, K = w.shape
        assert Cin_w == Cin, (w.shape, x.shape)
        assert Tin % stride == 0, "input length must be divisible by stride"   # blocks.py:46
        Tout = (Tin + 2 * (K // 2) - K) // stride + 1
        y = torch.empty((B, Tout, Cout), dtype=x.dtype, device=x.device)
        need_pre = act != ACT_NONE and any(ctx.needs_input_grad)
        ypre = torch.empty_like(y) if need_pre else None
        fl = 2.0 * B * Tout * Cout * Cin * K
        _timed(_conv_tag("fwd", K, stride, B, Tout, Cout), fl, lambda: check(
            lib.avdf_conv1d_fwd(_p(x), _p(w), _p(bias), _p(mask), _p(y), _p(ypre), B, Tin, Cin, Tout, Cout,
                                K, stride, act, _stream()), "conv1d_fwd"), ("fwd", K, stride, B, Tin, Cin, Cout, 1))
        ctx.save_for_backward(x, w, mask, ypre)
        ctx.cfg = (B, Tin, Cin, Tout, Cout, K, stride, act, bias is not None)
        ctx.main = (_main_grad(w0), _main_grad(b0))
        ctx.gmode = current_gemm_mode()
        return y

    @staticmethod
    def backward(ctx, dy):
        with _IssueMode(getattr(ctx, "gmode", 0)):
            return Conv1dFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, w, mask, ypre = ctx.saved_tensors
        B, Tin, Cin, Tout, Cout, K, stride, act, has_bias = ctx.cfg
        dy = dy.contiguous()
        dx = dw = db = None
        st = _stream()
        mw, mb = ctx.main
        if (K == 1 and stride == 1 and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and mw is not None
                and (not has_bias or mb is not None) and BWD_PAIR and not WGRAD_SIDE_STREAM):
            # engine path: dgrad + wgrad (accumulated into the flat gradient buffer) of the layer in one launch
            dx = torch.empty_like(x)
            fl = 4.0 * B * Tout * Cout * Cin
            _timed("conv_bwdpair_k1s1", fl, lambda: check(
                lib.avdf_conv1d_bwd_pair_grouped(1, _ptr_table([dy]), _ptr_table([w]), _ptr_table([x]),
                                                 _ptr_table([mask]) if mask is not None else None,
                                                 _ptr_table([ypre]) if ypre is not None else None, _ptr_table([dx]),
                                                 _ptr_table([mw.view(w.shape)]),
                                                 _ptr_table([mb.view(Cout)]) if has_bias else None,
                                                 B, Tin, Cin, Cout, act, st), "conv1d_bwd_pair"),
                   ("pair", K, stride, B, Tin, Cin, Cout, 1))
            return dx, None, None, None, None, None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            fl = 2.0 * B * Tout * Cout * Cin * K
            _timed(_conv_tag("dgrad", K, stride, B, Tin, Cin), fl, lambda: check(
                lib.avdf_conv1d_bwd_data(_p(dy), _p(w), _p(mask), _p(ypre), _p(dx), B, Tin, Cin, Tout, Cout,
                                         K, stride, act, st), "conv1d_bwd_data"), ("dgrad", K, stride, B, Tin, Cin, Cout, 1))
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            mw, mb = ctx.main
            acc = 1 if mw is not None else 0
            dw = mw.view(w.shape) if acc else torch.empty_like(w)
            db = None
            if has_bias:
                db = mb.view(Cout) if acc else torch.empty(Cout, dtype=w.dtype, device=w.device)
            fl = 2.0 * B * Tout * Cout * Cin * K
            with _WgradSide([dy, x, mask, ypre] if acc else [x[:0].cpu()]):
                ws = _ws(lib.avdf_conv1d_bwd_weight_ws(B, Tout, Cin, Cout, K), x)
                _timed(f"conv_wgrad_k{K}s{stride}", fl,
                       lambda: check(lib.avdf_conv1d_bwd_weight(_p(dy), _p(x), _p(mask), _p(ypre), _p(dw), _p(db), _p(ws),
                                                                B, Tin, Cin, Tout, Cout, K, stride, act, acc, _stream()),
                                     "conv1d_bwd_weight"), ("wgrad", K, stride, B, Tin, Cin, Cout, 1))
            if acc:
                dw = db = None
        return dx, dw, db, None, None, None


BWD_PAIR = True      # K=1 layers: data + weight gradient in one launch (avdf_conv1d_bwd_pair_grouped) on the engine path

# Arithmetic mode of the conv GEMMs, a bit mask: 0 = exact fp32 MFMA (the default EVERYWHERE: training, model.eval(), the
# extractors -- what every parity fixture is held against); bit 0 = split-bf16 operands ("bf16x3", ~5e-6 relative error,
# 1.3-1.9x faster) in the forward kernels; bit 1 = the same in the backward kernels of the layers whose forward was issued
# under it (data gradients, K = 1 weight gradients).  The mode is per THREAD on both sides of the C ABI (threading.local here, thread_local in the
# library), so an evaluation / prefetch thread cannot change the arithmetic of a training thread.
import threading as _threading

_tls = _threading.local()
_ENV_MODE = int(_os.environ.get("AVDF_GEMM_MODE", "0") or 0)
# opt-in for the inference-only paths (model.eval(), InferStep, the frozen extractors); AVDF_INFER_BF16X3=1 or assignment
INFER_BF16X3 = _os.environ.get("AVDF_INFER_BF16X3", "0") == "1"


def current_gemm_mode():
    return getattr(_tls, "mode", _ENV_MODE)


def set_gemm_mode(mode):
    """Arithmetic mode (bit mask, see above) of the conv GEMMs launched by THIS thread from now on (see avdf_gemm_mode)."""
    check(lib.avdf_gemm_mode(int(mode)), "gemm_mode")
    _tls.mode = int(mode)


class gemm_mode:
    """with ops.gemm_mode(1): ...   -- forward conv GEMMs launched inside (by this thread) use the given mode."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = current_gemm_mode()
        set_gemm_mode(self.mode)
        return self

    def __exit__(self, *a):
        set_gemm_mode(self.prev)
        return False


def infer_mode(bf16x3=None):
    """Arithmetic mode of the inference-only paths: exact fp32 unless bf16x3 is asked for (argument, `INFER_BF16X3`, or
    AVDF_INFER_BF16X3=1); an enclosing `gemm_mode(1)` is kept."""
    want = INFER_BF16X3 if bf16x3 is None else bool(bf16x3)
    return gemm_mode((current_gemm_mode() | 1) if want else current_gemm_mode())


class _IssueMode:
    """Backward launches run on the autograd worker thread, whose library mode is that thread's own default: carry the
    mode the layer's FORWARD was issued under (saved in ctx) across, so `gemm_mode(3)` around a forward pass selects the
    arithmetic of that pass's backward GEMMs (bit 1) as well, and nothing else in the process."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        self.prev = current_gemm_mode()
        if self.prev != self.mode:
            set_gemm_mode(self.mode)

    def __exit__(self, *a):
        if self.prev != self.mode:
            set_gemm_mode(self.prev)
        return False


class _ExactGemm:
    """Force the exact-fp32 forward GEMM for the launches inside the block (launch-time host switch of this thread)."""

    def __enter__(self):
        self.prev = current_gemm_mode()
        if self.prev != 0:
            set_gemm_mode(0)

    def __exit__(self, *a):
        if self.prev != 0:
            set_gemm_mode(self.prev)
        return False


def conv1d(x, w, bias=None, mask=None, stride=1, act=ACT_NONE, exact=False):
    """exact=True keeps the exact-fp32 MFMA even in bf16x3 mode: used where the output feeds an InstanceNorm over
    time, which can amplify rounding differences of near-constant channels by 1/sqrt(var + eps) ~ 300."""
    if exact:
        with _ExactGemm():
            return Conv1dFn.apply(x, w, bias, mask, stride, act)
    return Conv1dFn.apply(x, w, bias, mask, stride, act)


class _CtxProxy:
    """Duck-typed ctx so that a composite Function can reuse Conv1dFn.backward."""

    def __init__(self, saved, cfg, main, needs):
        self.saved_tensors, self.cfg, self.main, self.needs_input_grad = saved, cfg, main, needs


class Conv1dResidualFn(torch.autograd.Function):
    """out = a + scale[c] * keep[b] * ((conv1d(x, w) + bias) * mask): the MLP's second linear layer with the block's
    AffineDropPath + residual in the GEMM epilogue (blocks.py:1313).  Backward: da = dout; the gradient reaching the
    conv output and dscale come from the affine-residual backward kernel, then the ordinary conv backward."""

    @staticmethod
    def forward(ctx, x, w, bias, mask, a, scale, keep):
        w0, b0 = w, bias
        x, w, bias, mask, a, keep = _c(x), _c(w), _c(bias), _c(mask), _c(a), _c(keep)
        B, Tin, Cin = x.shape
        Cout, Cin_w, K = w.shape
        assert Cin_w == Cin and a.shape == (B, Tin, Cout), (w.shape, x.shape, a.shape)
        sc = scale.contiguous().view(-1)
        need_h = any(ctx.needs_input_grad)
        h = torch.empty((B, Tin, Cout), dtype=x.dtype, device=x.device) if need_h else None
        out = torch.empty((B, Tin, Cout), dtype=x.dtype, device=x.device)
        fl = 2.0 * B * Tin * Cout * Cin * K
        _timed(_conv_tag("fwd", K, 1, B, Tin, Cout), fl, lambda: check(
            lib.avdf_conv1d_fwd_residual(_p(x), _p(w), _p(bias), _p(mask), _p(a), _p(sc), _p(keep), _p(h), _p(out), B, Tin, Cin,
                                         Tin, Cout, K, 1, _stream()), "conv1d_fwd_residual"), ("fwd", K, 1, B, Tin, Cin, Cout, 1))
        ctx.save_for_backward(x, w, mask, h, sc, keep)
        ctx.cfg = (B, Tin, Cin, Tin, Cout, K, 1, ACT_NONE, bias is not None)
        ctx.main = (_main_grad(w0), _main_grad(b0))
        ctx.smain, ctx.sshape = _main_grad(scale), scale.shape
        ctx.gmode = current_gemm_mode()
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, mask, h, sc, keep = ctx.saved_tensors
        B, Tin, Cin, Tout, Cout, K, stride, act, has_bias = ctx.cfg
        dout = dout.contiguous()
        dh = torch.empty_like(h)
        acc = 1 if ctx.smain is not None else 0
        ds = None
        if ctx.needs_input_grad[5]:
            ds = ctx.smain.view(-1) if acc else torch.empty_like(sc)
        _affine_residual_bwd(dout, None, h, None, sc, keep, None, dh, ds, B, Tout, Cout, acc, h)
        proxy = _CtxProxy((x, w, mask, None), ctx.cfg, ctx.main, ctx.needs_input_grad[0:3] + (False,) * 3)
        proxy.gmode = getattr(ctx, "gmode", 0)
        dx, dw, db = Conv1dFn.backward(proxy, dh)[0:3]
        return dx, dw, db, None, dout, None if (ds is None or acc) else ds.view(ctx.sshape), None


FUSE_FC2_RESIDUAL = True   # blocks._residual_mlp: fc2 + AffineDropPath + residual as one launch
FUSE_MLP = _os.environ.get("AVDF_FUSE_MLP", "1") != "0"   # engine path: the whole MLP half of a block as one autograd node (MlpResidualFn)


class MlpResidualFn(torch.autograd.Function):
    """out = a + scale[c] * keep[b] * ((fc2(GELU(fc1(x))) + b2) * mask): the MLP half of a transformer block
    (blocks.py:1291-1297, 1313) as ONE autograd node on the engine path (every parameter has a `main_grad` view).
    Forward = the two launches of conv1d(act=GELU) + conv1d_residual.  Backward: AffineDropPath + residual, then the fc2
    data + weight gradient launch with the GELU derivative folded into its data-gradient epilogue
    (avdf_conv1d_bwd_pair_dact: gelu'(pre1) once per element of the [tokens, 4C] hidden gradient), then the fc1 launch on
    that tensor as it is.  The per-layer path evaluates gelu' in the prologue of fc1's backward GEMMs instead -- once per
    tile visit, six times per element for the exp10 shapes."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, mask, a, scale, keep):
        mains = [_main_grad(t) for t in (w1, b1, w2, b2)]
        x, w1c, b1c, w2c, b2c, mask, a, keep = _c(x), _c(w1), _c(b1), _c(w2), _c(b2), _c(mask), _c(a), _c(keep)
        B, T, C = x.shape
        H, Cout = w1c.shape[0], w2c.shape[0]
        assert w1c.shape == (H, C, 1) and w2c.shape == (Cout, H, 1) and a.shape == (B, T, Cout), (w1c.shape, w2c.shape, a.shape)
        sc = scale.contiguous().view(-1)
        pre1 = torch.empty((B, T, H), dtype=x.dtype, device=x.device)
        h1 = torch.empty_like(pre1)
        _timed(_conv_tag("fwd", 1, 1, B, T, H), 2.0 * B * T * H * C, lambda: check(
            lib.avdf_conv1d_fwd(_p(x), _p(w1c), _p(b1c), None, _p(h1), _p(pre1), B, T, C, T, H, 1, 1, ACT_GELU, _stream()),
            "conv1d_fwd"), ("fwd", 1, 1, B, T, C, H, 1))
        h2 = torch.empty((B, T, Cout), dtype=x.dtype, device=x.device)
        out = torch.empty_like(h2)
        _timed(_conv_tag("fwd", 1, 1, B, T, Cout), 2.0 * B * T * Cout * H, lambda: check(
            lib.avdf_conv1d_fwd_residual(_p(h1), _p(w2c), _p(b2c), _p(mask), _p(a), _p(sc), _p(keep), _p(h2), _p(out), B, T, H,
                                         T, Cout, 1, 1, _stream()), "conv1d_fwd_residual"), ("fwd", 1, 1, B, T, H, Cout, 1))
        ctx.save_for_backward(x, w1c, w2c, mask, pre1, h1, h2, sc, keep)
        ctx.dims = (B, T, C, H, Cout)
        ctx.mains = mains
        ctx.smain, ctx.sshape = _main_grad(scale), scale.shape
        ctx.gmode = current_gemm_mode()
        return out

    @staticmethod
    def backward(ctx, dout):
        with _IssueMode(getattr(ctx, "gmode", 0)):
            return MlpResidualFn._backward(ctx, dout)

    @staticmethod
    def _backward(ctx, dout):
        x, w1, w2, mask, pre1, h1, h2, sc, keep = ctx.saved_tensors
        B, T, C, H, Cout = ctx.dims
        mw1, mb1, mw2, mb2 = ctx.mains
        dout = dout.contiguous()
        st = _stream()
        dh2 = torch.empty_like(h2)
        acc = 1 if ctx.smain is not None else 0
        ds = None
        if ctx.needs_input_grad[7]:
            ds = ctx.smain.view(-1) if acc else torch.empty_like(sc)
        _affine_residual_bwd(dout, None, h2, None, sc, keep, None, dh2, ds, B, T, Cout, acc, h2)
        g1 = torch.empty_like(pre1)                 # d loss / d pre1 = (dh2 * mask) w2 * gelu'(pre1)
        _timed("conv_bwdpair_k1s1", 4.0 * B * T * Cout * H, lambda: check(
            lib.avdf_conv1d_bwd_pair_dact(1, _ptr_table([dh2]), _ptr_table([w2]), _ptr_table([h1]),
                                          _ptr_table([mask]) if mask is not None else None, None, _ptr_table([g1]),
                                          _ptr_table([mw2.view(w2.shape)]), _ptr_table([mb2.view(Cout)]),
                                          _ptr_table([pre1]), ACT_GELU, B, T, H, Cout, ACT_NONE, st), "conv1d_bwd_pair_dact"),
               ("pair", 1, 1, B, T, H, Cout, 1))
        dx = torch.empty_like(x)
        _timed("conv_bwdpair_k1s1", 4.0 * B * T * H * C, lambda: check(
            lib.avdf_conv1d_bwd_pair_grouped(1, _ptr_table([g1]), _ptr_table([w1]), _ptr_table([x]), None, None,
                                             _ptr_table([dx]), _ptr_table([mw1.view(w1.shape)]),
                                             _ptr_table([mb1.view(H)]), B, T, C, H, ACT_NONE, st), "conv1d_bwd_pair"),
               ("pair", 1, 1, B, T, C, H, 1))
        return dx, None, None, None, None, None, dout, None if (ds is None or acc) else ds.view(ctx.sshape), None


def mlp_residual_fusable(x, w1, b1, w2, b2, a):
    """MlpResidualFn applies on the engine path only: all four parameters accumulate into flat-buffer views, the block
    keeps its width (fc2 output = residual width) and the input needs a gradient."""
    return (FUSE_MLP and BWD_PAIR and not WGRAD_SIDE_STREAM and torch.is_grad_enabled() and x.requires_grad and b1 is not None
            and b2 is not None and all(_main_grad(t) is not None for t in (w1, b1, w2, b2)) and w2.shape[0] == a.shape[-1]
            and w1.shape[2] == 1 and w2.shape[2] == 1)


def mlp_residual(x, w1, b1, w2, b2, mask, a, scale, keep=None):
    return MlpResidualFn.apply(x, w1, b1, w2, b2, mask, a, scale, keep)


def conv1d_residual(x, w, bias, mask, a, scale, keep=None):
    return Conv1dResidualFn.apply(x, w, bias, mask, a, scale, keep)


class Conv1dGroup3Fn(torch.autograd.Function):
    """Three same-shape dense convs (no mask, no activation) as ONE launch each for forward, dgrad and wgrad:
    the q / k / v 1x1 projections (blocks.py:1176-1178, 734-736)."""

    @staticmethod
    def forward(ctx, x0, x1, x2, w0, w1, w2, b0, b1, b2):
        ws0, bs0 = [w0, w1, w2], [b0, b1, b2]
        xs, ws_, bs = [_c(t) for t in (x0, x1, x2)], [_c(t) for t in ws0], [_c(t) for t in bs0]
        B, T, Cin = xs[0].shape
        Cout, Cin_w, K = ws_[0].shape
        assert Cin_w == Cin and all(t.shape == xs[0].shape for t in xs) and all(t.shape == ws_[0].shape for t in ws_)
        has_bias = bs[0] is not None
        assert all((b is not None) == has_bias for b in bs)
        ys = [torch.empty((B, T, Cout), dtype=torch.float32, device=xs[0].device) for _ in range(3)]
        for t in xs + ws_ + ([b for b in bs] if has_bias else []):
            _p(t)
        fl = 3 * 2.0 * B * T * Cout * Cin * K
        _timed(_conv_tag("fwd3", K, 1, 3 * B, T, Cout), fl, lambda: check(
            lib.avdf_conv1d_fwd_grouped(3, _ptr_table(xs), _ptr_table(ws_), _ptr_table(bs) if has_bias else None, None,
                                        _ptr_table(ys), None, B, T, Cin, T, Cout, K, 1, ACT_NONE, _stream()),
            "conv1d_fwd_grouped"), ("fwd", K, 1, B, T, Cin, Cout, 3))
        ctx.save_for_backward(*xs, *ws_)
        ctx.cfg = (B, T, Cin, Cout, K, has_bias)
        ctx.main = ([_main_grad(t) for t in ws0], [_main_grad(t) for t in bs0])
        ctx.gmode = current_gemm_mode()
        return tuple(ys)

    @staticmethod
    def backward(ctx, d0, d1, d2):
        with _IssueMode(getattr(ctx, "gmode", 0)):
            return Conv1dGroup3Fn._backward(ctx, d0, d1, d2)

    @staticmethod
    def _backward(ctx, d0, d1, d2):
        sv = ctx.saved_tensors
        xs, ws_ = list(sv[0:3]), list(sv[3:6])
        B, T, Cin, Cout, K, has_bias = ctx.cfg
        dev = xs[0].device
        dys = [torch.zeros((B, T, Cout), dtype=torch.float32, device=dev) if d is None else d.contiguous() for d in (d0, d1, d2)]
        st = _stream()
        fl = 3 * 2.0 * B * T * Cout * Cin * K
        dxs = [None] * 3
        mw, mb = ctx.main
        acc = 1 if all(m is not None for m in mw) and (not has_bias or all(m is not None for m in mb)) else 0
        if K == 1 and acc and all(ctx.needs_input_grad[0:3]) and BWD_PAIR and not WGRAD_SIDE_STREAM:
            dxs = [torch.empty_like(x) for x in xs]
            _timed("conv_bwdpair3_k1s1", 2 * fl, lambda: check(
                lib.avdf_conv1d_bwd_pair_grouped(3, _ptr_table(dys), _ptr_table(ws_), _ptr_table(xs), None, None,
                                                 _ptr_table(dxs), _ptr_table([m.view(ws_[0].shape) for m in mw]),
                                                 _ptr_table([m.view(Cout) for m in mb]) if has_bias else None,
                                                 B, T, Cin, Cout, ACT_NONE, st), "conv1d_bwd_pair"),
                   ("pair", K, 1, B, T, Cin, Cout, 3))
            return (*dxs, None, None, None, None, None, None)
        if any(ctx.needs_input_grad[0:3]):
            dxs = [torch.empty_like(x) for x in xs]
            _timed(_conv_tag("dgrad3", K, 1, 3 * B, T, Cin), fl, lambda: check(
                lib.avdf_conv1d_bwd_data_grouped(3, _ptr_table(dys), _ptr_table(ws_), None, None, _ptr_table(dxs), B, T, Cin,
                                                 T, Cout, K, 1, ACT_NONE, st), "conv1d_bwd_data_grouped"),
                   ("dgrad", K, 1, B, T, Cin, Cout, 3))
        mw, mb = ctx.main
        acc = 1 if all(m is not None for m in mw) and (not has_bias or all(m is not None for m in mb)) else 0
        dws = [m.view(ws_[0].shape) for m in mw] if acc else [torch.empty_like(w) for w in ws_]
        dbs = None
        if has_bias:
            dbs = [m.view(Cout) for m in mb] if acc else [torch.empty(Cout, dtype=torch.float32, device=dev) for _ in range(3)]
        per = (int(lib.avdf_conv1d_bwd_weight_ws(B, T, Cin, Cout, K)) + 3) // 4 * 4
        with _WgradSide(dys + xs if acc else [xs[0][:0].cpu()]):
            wsb = _ws(3 * per, xs[0])
            _timed(f"conv_wgrad3_k{K}s1", fl, lambda: check(
                lib.avdf_conv1d_bwd_weight_grouped(3, _ptr_table(dys), _ptr_table(xs), None, None, _ptr_table(dws),
                                                   _ptr_table(dbs) if has_bias else None, _p(wsb), B, T, Cin, T, Cout, K, 1,
                                                   ACT_NONE, acc, _stream()), "conv1d_bwd_weight_grouped"),
                   ("wgrad", K, 1, B, T, Cin, Cout, 3))
        if acc:
            return (*dxs, None, None, None, None, None, None)
        return (*dxs, *dws, *(dbs if has_bias else [None] * 3))


def conv1d_group3(xs, ws, bs):
    return Conv1dGroup3Fn.apply(xs[0], xs[1], xs[2], ws[0], ws[1], ws[2], bs[0], bs[1], bs[2])


class ConvTranspose1dFn(torch.autograd.Function):
    """nn.ConvTranspose1d(k=3, stride=2, padding=1, output_padding=1) on time-major tensors + bias + mask:
    y[b, t', co] = (sum_{ci,k} x[b, t, ci] * w[ci, co, k] + bias[co]) * mask[b, t'],  t' = 2t - 1 + k.
    This IS the data-gradient of a stride-2 conv whose weight is w read as [Cout_conv = ci, Cin_conv = co, K],
    so forward = avdf_conv1d_bwd_data, d/dx = avdf_conv1d_fwd and d/dw = avdf_conv1d_bwd_weight with the roles
    of activation and gradient swapped (MaskedConvTranspose1D, blocks.py:1443-1491)."""

    @staticmethod
    def forward(ctx, x, w, bias, mask_out):
        ctx.main = (_main_grad(w), _main_grad(bias))
        x, w, bias, mask_out = _c(x), _c(w), _c(bias), _c(mask_out)
        B, T, Ci = x.shape
        Ci_w, Co, K = w.shape
        assert Ci_w == Ci and K == 3
        T2 = 2 * T
        st = _stream()
        raw = torch.empty((B, T2, Co), dtype=x.dtype, device=x.device)
        ctx.gmode = current_gemm_mode()
        with _IssueMode(2 if (ctx.gmode & 1) else 0):      # the forward pass IS a data-gradient kernel: swap the mode bits
            check(lib.avdf_conv1d_bwd_data(_p(x), _p(w), None, None, _p(raw), B, T2, Co, T, Ci, 3, 2, ACT_NONE, st),
                  "conv_transpose1d_fwd")
        y = torch.empty_like(raw)
        check(lib.avdf_bias_mask_fwd(_p(raw), _p(bias), _p(mask_out), _p(y), B * T2, Co, st), "bias_mask_fwd")
        ctx.save_for_backward(x, w, mask_out)
        ctx.cfg = (B, T, Ci, Co, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mask_out = ctx.saved_tensors
        B, T, Ci, Co, has_bias = ctx.cfg
        T2 = 2 * T
        st = _stream()
        g = torch.empty_like(dy)                                   # d/d(raw) = dy * mask
        check(lib.avdf_bias_mask_fwd(_p(dy.contiguous()), None, _p(mask_out), _p(g), B * T2, Co, st), "bias_mask_bwd")
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with _IssueMode(1 if (getattr(ctx, "gmode", 0) & 2) else 0):
                check(lib.avdf_conv1d_fwd(_p(g), _p(w), None, None, _p(dx), None, B, T2, Co, T, Ci, 3, 2, ACT_NONE, st),
                      "conv_transpose1d_dx")
        mw, mb = ctx.main
        acc = 1 if mw is not None else 0
        if ctx.needs_input_grad[1]:
            dw = mw.view(w.shape) if acc else torch.empty_like(w)
            ws = _ws(lib.avdf_conv1d_bwd_weight_ws(B, T, Co, Ci, 3), x)
            with _IssueMode(0):
                check(lib.avdf_conv1d_bwd_weight(_p(x), _p(g), None, None, _p(dw), None, _p(ws), B, T2, Co, T, Ci, 3, 2,
                                                 ACT_NONE, acc, st), "conv_transpose1d_dw")
        if has_bias and ctx.needs_input_grad[2]:
            accb = 1 if mb is not None else 0
            db = mb.view(Co) if accb else torch.empty(Co, dtype=w.dtype, device=w.device)
            ws2 = _ws(lib.avdf_masked_colsum_ws(Co), x)
            check(lib.avdf_masked_colsum(_p(g), None, _p(db), _p(ws2), B * T2, Co, accb, st), "masked_colsum")
            if accb:
                db = None
        if acc:
            dw = None
        return dx, dw, db, None


def conv_transpose1d_k3s2(x, w, bias, mask_out):
    return ConvTranspose1dFn.apply(x, w, bias, mask_out)


class L1MeanFn(torch.autograd.Function):
    """mean(|a[sel] - b[sel]|) over the clips with wb == 1 (b is a constant target)."""

    @staticmethod
    def forward(ctx, a, b, wb, n_sel):
        a, b, wb = _c(a), _c(b), _c(wb)
        B = a.shape[0]
        per = a.numel() // B
        denom = float(n_sel) * per
        inv = (1.0 / denom) if denom > 0 else float("nan")          # torch.mean of an empty tensor is nan
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        ws = _ws(lib.avdf_l1_mean_ws(), a)
        check(lib.avdf_l1_mean(_p(a), _p(b), _p(wb), _p(out), _p(da), _p(ws), B, per, inv, _stream()), "l1_mean")
        ctx.save_for_backward(da)
        return out

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        return da * g, None, None, None


def l1_mean(a, b, wb, n_sel):
    return L1MeanFn.apply(a, b, wb, n_sel)


def linear(x2d, w, bias=None, act=ACT_NONE):
    """nn.Linear on [M, Cin] rows via the K=1 conv path (weight [out, in])."""
    M, Cin = x2d.shape
    y = conv1d(x2d.view(1, M, Cin), view_param(w, (w.shape[0], Cin, 1)), bias, None, 1, act)
    return y.view(M, w.shape[0])


def view_param(p, shape):
    """Reshaped view of a parameter that keeps its `main_grad` target (see _main_grad)."""
    v = p.view(shape)
    mg = getattr(p, "main_grad", None)
    if mg is not None:
        v.main_grad = mg
    return v


# --------------------------------------------------------------------------------------------
# LayerNorm over channels
# --------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, pe, mask, rowscale=None):
        x = _c(x)
        shp = x.shape
        C = shp[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        need = any(ctx.needs_input_grad)
        mean = torch.empty(M, dtype=x.dtype, device=x.device) if need else None
        rstd = torch.empty(M, dtype=x.dtype, device=x.device) if need else None
        T = shp[-2] if pe is not None else 0
        g = None if gamma is None else gamma.contiguous().view(-1)
        b = None if beta is None else beta.contiguous().view(-1)
        rs = None
        if rowscale is not None:
            # y = act(LN(x)) * rowscale[row]: the heads zero the gap rows of the concatenated level sequence in the LN launch
            assert pe is None and mask is None and rowscale.numel() == M
            rs = rowscale.contiguous().view(-1)
            check(lib.avdf_layernorm_fwd_rs(_p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), M, C, eps, int(relu), _p(rs),
                                            _stream()), "layernorm_fwd_rs")
        else:
            check(lib.avdf_layernorm_fwd(_p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), M, C, eps, int(relu),
                                         _p(_c(pe)), _p(_c(mask)), T, _stream()), "layernorm_fwd")
        ctx.rs = rs
        ctx.save_for_backward(x, g, b, mean, rstd)
        ctx.cfg = (M, C, int(relu), None if gamma is None else gamma.shape)
        ctx.main = (_main_grad(gamma), _main_grad(beta))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g, b, mean, rstd = ctx.saved_tensors
        M, C, relu, gshape = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        mg, mb = ctx.main
        acc = 1 if (mg is not None and mb is not None) else 0
        dg = db = None
        if g is not None:
            dg = mg.view(C) if acc else torch.empty(C, dtype=x.dtype, device=x.device)
            db = mb.view(C) if acc else torch.empty(C, dtype=x.dtype, device=x.device)
        nws = int(lib.avdf_layernorm_bwd_ws(M, C))
        ws = _ws(nws, x)
        defer = bool(acc) and dg is not None and DEFER_ROW_PARTIALS and _deferrable(2 * C, [C, C])
        mode = 2 if defer else acc
        if ctx.rs is not None:
            check(lib.avdf_layernorm_bwd_rs(_p(dy), _p(x), _p(g), _p(b), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), _p(ws),
                                            M, C, relu, mode, _p(ctx.rs), _stream()), "layernorm_bwd_rs")
        else:
            check(lib.avdf_layernorm_bwd(_p(dy), _p(x), _p(g), _p(b), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), _p(ws),
                                         M, C, relu, mode, _stream()), "layernorm_bwd")
        if defer:
            _DEFERRED.append((ws, nws // (2 * C), 2 * C, [dg, db], [C, C], [[0], [C]]))
        if acc:
            dg = db = None
        elif dg is not None:
            dg, db = dg.view(gshape), db.view(gshape)
        return dx, dg, db, None, None, None, None, None


def layernorm(x, gamma, beta, eps=1e-5, relu=False, pe=None, mask=None, rowscale=None):
    return LayerNormFn.apply(x, gamma, beta, eps, relu, pe, mask, rowscale)


# --------------------------------------------------------------------------------------------
# fused q/k/v input chains: LN -> depthwise k3 (stride s) * mask -> LN, three branches per launch
# --------------------------------------------------------------------------------------------
class QkvChainFn(torch.autograd.Function):
    """args: x[3], (gamma_a, beta_a)[3], w[3], (gamma_b, beta_b)[3], mask_out, stride, eps -> y[3].
    Inputs / parameters of different branches may be the same tensor (TransformerBlock: one x and one ln1)."""

    @staticmethod
    def forward(ctx, xq, xk, xv, gaq, baq, gak, bak, gav, bav, wq, wk, wv, gbq, bbq, gbk, bbk, gbv, bbv, mask, stride,
                eps, passthrough=False, xshift=None):
        # xshift (optional, per branch): nearest resampling of that branch's input folded into the kernels' row address
        # (> 0: the input has Tin >> s rows and is read as if up-sampled by 2^s; < 0: Tin << -s rows, every 2^-s-th read)
        xshift = tuple(int(v) for v in (xshift or (0, 0, 0)))
        xs = [_c(xq), _c(xk), _c(xv)]
        ga, ba = [gaq, gak, gav], [baq, bak, bav]
        ws_, gb, bb = [wq, wk, wv], [gbq, gbk, gbv], [bbq, bbk, bbv]
        ctx.main = [[_main_grad(t) for t in grp] for grp in (ga, ba, ws_, gb, bb)]
        ctx.pshapes = [[t.shape for t in grp] for grp in (ga, ba, ws_, gb, bb)]
        flat = lambda grp: [t.contiguous().view(-1) for t in grp]
        ga, ba, ws_, gb, bb = flat(ga), flat(ba), flat(ws_), flat(gb), flat(bb)
        mask = _c(mask)
        B, Tin, C = xs[0].shape
        assert xshift[0] == 0, "the query branch defines the chain's resolution"
        for xi, sh in zip(xs, xshift):
            assert xi.shape == (B, (Tin >> sh) if sh > 0 else (Tin << -sh), C), (xi.shape, (B, Tin, C), sh)
        Tout = (Tin - 1) // stride + 1
        dev = xs[0].device
        u = [torch.empty((B, Tout, C), dtype=torch.float32, device=dev) for _ in range(3)]
        y = [torch.empty((B, Tout, C), dtype=torch.float32, device=dev) for _ in range(3)]
        sa = torch.empty((3, 2, B * Tin), dtype=torch.float32, device=dev)
        sb = torch.empty((3, 2, B * Tout), dtype=torch.float32, device=dev)
        for t in xs + ga + ba + ws_ + gb + bb:
            _p(t)
        if any(xshift):
            check(lib.avdf_chain_fwd_map(_ptr_table(xs), _ptr_table(ga), _ptr_table(ba), _ptr_table(ws_), _ptr_table(gb),
                                         _ptr_table(bb), _p(mask), _ptr_table(u), _ptr_table(y),
                                         _ptr_table([sa[0], sa[1], sa[2]]), _ptr_table([sb[0], sb[1], sb[2]]),
                                         (ctypes.c_int * 3)(*xshift), B, Tin, Tout, C, stride, eps, _stream()), "chain_fwd_map")
        else:
            check(lib.avdf_chain_fwd(_ptr_table(xs), _ptr_table(ga), _ptr_table(ba), _ptr_table(ws_), _ptr_table(gb),
                                     _ptr_table(bb), _p(mask), _ptr_table(u), _ptr_table(y), _ptr_table([sa[0], sa[1], sa[2]]),
                                     _ptr_table([sb[0], sb[1], sb[2]]), B, Tin, Tout, C, stride, eps, _stream()), "chain_fwd")
        ctx.save_for_backward(*xs, *ga, *ba, *ws_, *gb, mask, *u, sa, sb)
        ctx.cfg = (B, Tin, Tout, C, stride)
        ctx.xshift = xshift
        ctx.x_same = [xs[i + 1].data_ptr() == xs[i].data_ptr() for i in range(2)]
        assert all(xshift[i] == xshift[i + 1] for i in range(2) if ctx.x_same[i]), "one input, two different row maps"
        ctx.passthrough = bool(passthrough)
        if ctx.passthrough:
            # 4th output = x_q itself: the caller uses it for the block's residual connection, so x_q has ONE consumer
            # in the autograd graph and the residual gradient arrives here (added into dx by the backward kernel)
            return (*y, xq)
        return tuple(y)

    @staticmethod
    def backward(ctx, dyq, dyk, dyv, dres=None):
        sv = ctx.saved_tensors
        xs, ga, ba, ws_, gb = list(sv[0:3]), list(sv[3:6]), list(sv[6:9]), list(sv[9:12]), list(sv[12:15])
        mask, u, sa, sb = sv[15], list(sv[16:19]), sv[19], sv[20]
        B, Tin, Tout, C, stride = ctx.cfg
        dev = xs[0].device
        dys = [torch.zeros((B, Tout, C), dtype=torch.float32, device=dev) if d is None else d.contiguous()
               for d in (dyq, dyk, dyv)]
        # branches whose input is the same tensor are summed in the kernel; the sum is stored by the group's last branch
        store = [not ctx.x_same[0], not ctx.x_same[1], True]
        # (an up-sampled input whose row groups the kernel sums itself gets its gradient at the SOURCE resolution)
        folds = [bool(sh > 0 and lib.avdf_chain_bwd_folds(sh, stride)) for sh in ctx.xshift]
        dx = [(torch.empty_like(xs[i]) if folds[i] else torch.empty((B, Tin, C), dtype=torch.float32, device=dev)) if st else None
              for i, st in enumerate(store)]
        store_mask = sum(1 << i for i, st in enumerate(store) if st)
        wsb = _ws(lib.avdf_chain_bwd_ws(B, Tin, C), xs[0])
        res = [None, None, None]
        if dres is not None:
            # the branch that closes x_q's input group stores dx for it
            res[2 if (ctx.x_same[0] and ctx.x_same[1]) else (1 if ctx.x_same[0] else 0)] = dres.contiguous()
        if any(ctx.xshift):
            check(lib.avdf_chain_bwd_map(_ptr_table(dys), _ptr_table(u), _ptr_table([sb[0], sb[1], sb[2]]), _ptr_table(xs),
                                         _ptr_table([sa[0], sa[1], sa[2]]), _ptr_table(ga), _ptr_table(ba), _ptr_table(ws_),
                                         _ptr_table(gb), _p(mask), _ptr_table(dx), _ptr_table(res), store_mask, _p(wsb),
                                         (ctypes.c_int * 3)(*ctx.xshift), B, Tin, Tout, C, stride, _stream()), "chain_bwd_map")
            # dx of a resampled input is at the chain's resolution: fold it back to the source rows (the backward of the
            # nearest up / down-sampling that the forward skipped)
            for i, sh in enumerate(ctx.xshift):
                if sh == 0 or dx[i] is None or folds[i]:
                    continue
                src = torch.empty_like(xs[i])
                if sh > 0:
                    check(lib.avdf_upsample_bwd(_p(dx[i]), _p(src), B, Tin >> sh, C, 1 << sh, _stream()), "upsample_bwd")
                else:
                    check(lib.avdf_downsample_bwd(_p(dx[i]), _p(src), B, Tin << -sh, C, 1 << -sh, _stream()), "downsample_bwd")
                dx[i] = src
        else:
            check(lib.avdf_chain_bwd_res(_ptr_table(dys), _ptr_table(u), _ptr_table([sb[0], sb[1], sb[2]]), _ptr_table(xs),
                                         _ptr_table([sa[0], sa[1], sa[2]]), _ptr_table(ga), _ptr_table(ba), _ptr_table(ws_),
                                         _ptr_table(gb), _p(mask), _ptr_table(dx), _ptr_table(res), store_mask, _p(wsb), B, Tin,
                                         Tout, C, stride, _stream()), "chain_bwd")
        # parameter gradients: one segmented reduction over the workgroups' rows; sources of a parameter shared by
        # several branches are summed there
        mains = ctx.main
        acc = all(m is not None for grp in mains for m in grp)
        slot = {0: 0, 1: C, 2: 2 * C, 3: 5 * C, 4: 6 * C}
        seg_dst, seg_len, seg_src, grads = [], [], [], [[None] * 3 for _ in range(5)]
        params = (ga, ba, ws_, gb, sv[12:15])      # beta_b is not saved (not needed); its identity follows gamma_b
        for kind in range(5):
            seen = {}
            for br in range(3):
                key = params[kind][br].data_ptr() if kind != 4 else ("bb", params[3][br].data_ptr())
                off = br * 7 * C + slot[kind]
                if key in seen:
                    seg_src[seen[key]].append(off)
                    continue
                n = 3 * C if kind == 2 else C
                if acc:
                    dst = mains[kind][br].view(-1)
                else:
                    dst = torch.empty(n, dtype=torch.float32, device=dev)
                    grads[kind][br] = dst.view(ctx.pshapes[kind][br])
                seen[key] = len(seg_dst)
                seg_dst.append(dst)
                seg_len.append(n)
                seg_src.append([off])
        nseg = len(seg_dst)
        lens = (ctypes.c_int * nseg)(*seg_len)
        nsrc = (ctypes.c_int * nseg)(*[len(s_) for s_ in seg_src])
        offs = (ctypes.c_int * (nseg * 3))(*[(s_ + [0, 0, 0])[j] for s_ in seg_src for j in range(3)])
        if acc and _deferrable(21 * C, seg_len, seg_src):
            # engine path: folded together with the other chains of this backward phase (flush_deferred_reductions)
            _DEFERRED.append((wsb, int(lib.avdf_chain_bwd_rows(B, Tin)), 21 * C, seg_dst, seg_len, seg_src))
        else:
            check(lib.avdf_reduce_segments(_p(wsb), lib.avdf_chain_bwd_rows(B, Tin), 21 * C, nseg, _ptr_table(seg_dst), lens, nsrc,
                                           offs, 1 if acc else 0, _stream()), "reduce_segments")
        g = grads
        return (dx[0], dx[1], dx[2], g[0][0], g[1][0], g[0][1], g[1][1], g[0][2], g[1][2], g[2][0], g[2][1], g[2][2],
                g[3][0], g[4][0], g[3][1], g[4][1], g[3][2], g[4][2], None, None, None, None, None)


def qkv_chain(xs, ln_a, dw, ln_b, mask_out, stride, eps=1e-5, passthrough=False, xshift=None):
    """xs: 3 inputs [B,T,C]; ln_a / ln_b: 3 (gamma, beta) pairs; dw: 3 depthwise weights [C,1,3] -> 3 outputs
    (+ xs[0] handed back as a 4th output when `passthrough`: use THAT tensor for the residual connection and the
    gradient of the residual path is summed inside the chain's backward kernel instead of by a separate add)."""
    return QkvChainFn.apply(xs[0], xs[1], xs[2], ln_a[0][0], ln_a[0][1], ln_a[1][0], ln_a[1][1], ln_a[2][0], ln_a[2][1],
                            dw[0], dw[1], dw[2], ln_b[0][0], ln_b[0][1], ln_b[1][0], ln_b[1][1], ln_b[2][0], ln_b[2][1],
                            mask_out, stride, eps, passthrough, xshift)


# --------------------------------------------------------------------------------------------
# depthwise conv k3
# --------------------------------------------------------------------------------------------
class DwConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, mask, stride):
        ctx.main = _main_grad(w)
        x, w, mask = _c(x), _c(w), _c(mask)
        B, Tin, C = x.shape
        assert w.shape == (C, 1, 3), w.shape
        assert Tin % stride == 0, "input length must be divisible by stride"
        Tout = (Tin - 1) // stride + 1
        y = torch.empty((B, Tout, C), dtype=x.dtype, device=x.device)
        check(lib.avdf_dwconv_fwd(_p(x), _p(w), _p(mask), _p(y), B, Tin, Tout, C, stride, _stream()), "dwconv_fwd")
        ctx.save_for_backward(x, w, mask)
        ctx.cfg = (B, Tin, Tout, C, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mask = ctx.saved_tensors
        B, Tin, Tout, C, stride = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        acc = 1 if ctx.main is not None else 0
        dw = None
        if ctx.needs_input_grad[1]:
            dw = ctx.main.view(w.shape) if acc else torch.empty_like(w)
        ws = _ws(lib.avdf_dwconv_bwd_ws(B, Tout, C), x) if dw is not None else None
        check(lib.avdf_dwconv_bwd(_p(dy), _p(x), _p(w), _p(mask), _p(dx), _p(dw), _p(ws), B, Tin, Tout, C, stride, acc,
                                  _stream()), "dwconv_bwd")
        return dx, (None if acc else dw), None, None


def dwconv(x, w, mask=None, stride=1):
    return DwConvFn.apply(x, w, mask, stride)


# --------------------------------------------------------------------------------------------
# attention cores
# --------------------------------------------------------------------------------------------
class LocalAttnFn(torch.autograd.Function):
    """Banded attention core; rel_pe [1,1,H,W] / time_w [1,H,W] are the optional per-head tables of use_rel_pe /
    use_time_weight (blocks.py:968-975): a bias on the scores before the softmax, a weight on the probabilities after it."""

    @staticmethod
    def forward(ctx, q, k, v, kvmask, n_head, window, rel_pe=None, time_w=None):
        q, k, v, kvmask = _c(q), _c(k), _c(v), _c(kvmask)
        B, T, C = q.shape
        assert k.shape == q.shape and v.shape == q.shape, "local attention needs equal q/k/v lengths"
        assert T % (2 * (window // 2)) == 0      # blocks.py:1060
        hs = C // n_head
        scale = 1.0 / (hs ** 0.5)
        out = torch.empty_like(q)
        need = any(ctx.needs_input_grad)
        probs = torch.empty((B, T, n_head, window), dtype=q.dtype, device=q.device) if need else None
        rel = None if rel_pe is None else rel_pe.contiguous().view(-1)
        tw = None if time_w is None else time_w.contiguous().view(-1)
        assert (rel is None or rel.numel() == n_head * window) and (tw is None or tw.numel() == n_head * window)
        check(lib.avdf_local_attn_fwd_ex(_p(q), _p(k), _p(v), _p(kvmask), _p(rel), _p(tw), _p(out), _p(probs), B, T, n_head, hs,
                                         window, scale, _stream()), "local_attn_fwd")
        ctx.save_for_backward(q, k, v, probs, tw)
        ctx.cfg = (B, T, n_head, hs, window, scale)
        ctx.main = (_main_grad(rel_pe), _main_grad(time_w))
        ctx.shapes = (None if rel_pe is None else rel_pe.shape, None if time_w is None else time_w.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, probs, tw = ctx.saved_tensors
        B, T, H, hs, window, scale = ctx.cfg
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = _ws(lib.avdf_local_attn_bwd_ws(B, T, H, window), q)
        need_rel, need_tw = ctx.shapes[0] is not None, ctx.shapes[1] is not None
        dT = torch.empty(B * T * H * window, dtype=torch.float32, device=q.device) if need_tw else None
        check(lib.avdf_local_attn_bwd_ex(_p(dout), _p(q), _p(k), _p(v), _p(probs), _p(tw), _p(dq), _p(dk), _p(dv), _p(ws), _p(dT),
                                         B, T, H, hs, window, scale, _stream()), "local_attn_bwd")
        grads = [None, None]
        for i, (need, src) in enumerate(((need_rel, ws), (need_tw, dT))):
            if not need:
                continue
            # column sums over the B*T rows of the [B*T, H*window] summand matrix
            n = H * window
            main = ctx.main[i]
            acc = 1 if main is not None else 0
            dst = main.view(-1) if acc else torch.empty(n, dtype=torch.float32, device=q.device)
            ws2 = _ws(lib.avdf_masked_colsum_ws(n), q)
            check(lib.avdf_masked_colsum(_p(src), None, _p(dst), _p(ws2), B * T, n, acc, _stream()), "masked_colsum")
            grads[i] = None if acc else dst.view(ctx.shapes[i])
        return dq, dk, dv, None, None, None, grads[0], grads[1]


class GlobalAttnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kvmask, n_head):
        q, k, v, kvmask = _c(q), _c(k), _c(v), _c(kvmask)
        B, Tq, C = q.shape
        Tk = k.shape[1]
        hs = C // n_head
        scale = 1.0 / (hs ** 0.5)
        out = torch.empty_like(q)
        need = any(ctx.needs_input_grad)
        probs = torch.empty((B, n_head, Tq, Tk), dtype=q.dtype, device=q.device) if need else None
        check(lib.avdf_global_attn_fwd(_p(q), _p(k), _p(v), _p(kvmask), _p(out), _p(probs), B, Tq, Tk, n_head, hs,
                                       scale, _stream()), "global_attn_fwd")
        ctx.save_for_backward(q, k, v, kvmask, probs)
        ctx.cfg = (B, Tq, Tk, n_head, hs, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, kvmask, probs = ctx.saved_tensors
        B, Tq, Tk, H, hs, scale = ctx.cfg
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = _ws(lib.avdf_global_attn_bwd_ws(B, Tq, Tk, H), q)
        check(lib.avdf_global_attn_bwd(_p(dout), _p(q), _p(k), _p(v), _p(kvmask), _p(probs), _p(dq), _p(dk), _p(dv),
                                       _p(ws), B, Tq, Tk, H, hs, scale, _stream()), "global_attn_bwd")
        return dq, dk, dv, None, None


def local_attention(q, k, v, kvmask, n_head, window, rel_pe=None, time_w=None):
    return LocalAttnFn.apply(q, k, v, kvmask, n_head, window, rel_pe, time_w)


def global_attention(q, k, v, kvmask, n_head):
    return GlobalAttnFn.apply(q, k, v, kvmask, n_head)


# --------------------------------------------------------------------------------------------
# block glue
# --------------------------------------------------------------------------------------------
class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, Tin, C = x.shape
        y = torch.empty((B, (Tin - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
        check(lib.avdf_maxpool3s2_fwd(_p(x), _p(y), B, Tin, C, _stream()), "maxpool_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, Tin, C = x.shape
        dx = torch.empty_like(x)
        check(lib.avdf_maxpool3s2_bwd(_p(dy.contiguous()), _p(x), _p(dx), B, Tin, C, _stream()), "maxpool_bwd")
        return dx


def maxpool3s2(x):
    return MaxPoolFn.apply(x)


class AffineResidualFn(torch.autograd.Function):
    """out = a*ma + scale[c]*keep[b]*(r*mr)"""

    @staticmethod
    def forward(ctx, a, ma, r, mr, scale, keep):
        ctx.main = _main_grad(scale)
        a, ma, r, mr, keep = _c(a), _c(ma), _c(r), _c(mr), _c(keep)
        B, T, C = a.shape
        sc = scale.contiguous().view(-1)
        out = torch.empty_like(a)
        check(lib.avdf_affine_residual_fwd(_p(a), _p(ma), _p(r), _p(mr), _p(sc), _p(keep), _p(out), B, T, C,
                                           _stream()), "affine_residual_fwd")
        ctx.save_for_backward(ma, r, mr, sc, keep)
        ctx.sshape = scale.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        ma, r, mr, sc, keep = ctx.saved_tensors
        B, T, C = r.shape
        dout = dout.contiguous()
        da = torch.empty_like(r) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(r) if ctx.needs_input_grad[2] else None
        acc = 1 if ctx.main is not None else 0
        ds = None
        if ctx.needs_input_grad[4]:
            ds = ctx.main.view(-1) if acc else torch.empty_like(sc)
        _affine_residual_bwd(dout, ma, r, mr, sc, keep, da, dr, ds, B, T, C, acc, r)
        return da, None, dr, None, None if (ds is None or acc) else ds.view(ctx.sshape), None


def affine_residual(a, ma, r, mr, scale, keep=None):
    return AffineResidualFn.apply(a, ma, r, mr, scale, keep)


class ResidualLnFn(torch.autograd.Function):
    """(out, h) = (a*ma + scale*keep*(r*mr), LN(out)); `out` keeps flowing to the second residual of the block, so
    its incoming gradient and LN's are combined in ONE backward kernel (no aten add, no separate LN backward)."""

    @staticmethod
    def forward(ctx, a, ma, r, mr, scale, keep, gamma, beta, eps):
        ctx.main = (_main_grad(scale), _main_grad(gamma), _main_grad(beta))
        ctx.shapes = (scale.shape, gamma.shape, beta.shape)
        a, ma, r, mr, keep = _c(a), _c(ma), _c(r), _c(mr), _c(keep)
        B, T, C = a.shape
        sc, g, b = scale.contiguous().view(-1), gamma.contiguous().view(-1), beta.contiguous().view(-1)
        out, h = torch.empty_like(a), torch.empty_like(a)
        stats = torch.empty((2, B * T), dtype=torch.float32, device=a.device)
        check(lib.avdf_residual_ln_fwd(_p(a), _p(ma), _p(r), _p(mr), _p(sc), _p(keep), _p(g), _p(b), _p(out), _p(h),
                                       _p(stats[0]), _p(stats[1]), B, T, C, eps, _stream()), "residual_ln_fwd")
        ctx.save_for_backward(out, stats, g, ma, r, mr, sc, keep)
        return out, h

    @staticmethod
    def backward(ctx, dout, dh):
        out, stats, g, ma, r, mr, sc, keep = ctx.saved_tensors
        B, T, C = out.shape
        dev = out.device
        dout = None if dout is None else dout.contiguous()
        dh = torch.zeros_like(out) if dh is None else dh.contiguous()
        da = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(out) if ctx.needs_input_grad[2] else None
        ms, mg, mb = ctx.main
        acc = 1 if (ms is not None and mg is not None and mb is not None) else 0
        if acc:
            ds, dg, db = ms.view(-1), mg.view(-1), mb.view(-1)
        else:
            ds, dg, db = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(3))
        nws = int(lib.avdf_residual_ln_bwd_ws(B, T, C))
        ws = _ws(nws, out)
        defer = bool(acc) and DEFER_ROW_PARTIALS and _deferrable(3 * C, [C, C, C])
        check(lib.avdf_residual_ln_bwd(_p(dout), _p(dh), _p(out), _p(stats[0]), _p(stats[1]), _p(g), _p(ma), _p(r), _p(mr),
                                       _p(sc), _p(keep), _p(da), _p(dr), _p(ds), _p(dg), _p(db), _p(ws), B, T, C,
                                       2 if defer else acc, _stream()), "residual_ln_bwd")
        if defer:
            # engine path: the workgroups' partial rows are folded with the other parameter partials of this backward phase
            # (no float atomics inside the kernel: 128 workgroups x 768 adds onto the same 768 floats)
            _DEFERRED.append((ws, nws // (3 * C), 3 * C, [ds, dg, db], [C, C, C], [[0], [C], [2 * C]]))
        if acc:
            return da, None, dr, None, None, None, None, None, None
        s0, s1, s2 = ctx.shapes
        return da, None, dr, None, ds.view(s0), None, dg.view(s1), db.view(s2), None


def residual_ln(a, ma, r, mr, scale, keep, gamma, beta, eps=1e-5):
    return ResidualLnFn.apply(a, ma, r, mr, scale, keep, gamma, beta, eps)


class ResampleFn(torch.autograd.Function):
    """mode 'up': y[b,t]=x[b,t//r]; mode 'down': y[b,t]=x[b,t*r] (F.interpolate nearest)."""

    @staticmethod
    def forward(ctx, x, r, mode):
        x = _c(x)
        B, Tin, C = x.shape
        ctx.cfg = (B, Tin, C, r, mode)
        if mode == "up":
            y = torch.empty((B, Tin * r, C), dtype=x.dtype, device=x.device)
            check(lib.avdf_upsample_fwd(_p(x), _p(y), B, Tin, C, r, _stream()), "upsample_fwd")
        else:
            y = torch.empty((B, Tin // r, C), dtype=x.dtype, device=x.device)
            check(lib.avdf_downsample_fwd(_p(x), _p(y), B, Tin, C, r, _stream()), "downsample_fwd")
        return y

    @staticmethod
    def backward(ctx, dy):
        B, Tin, C, r, mode = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((B, Tin, C), dtype=dy.dtype, device=dy.device)
        if mode == "up":
            check(lib.avdf_upsample_bwd(_p(dy), _p(dx), B, Tin, C, r, _stream()), "upsample_bwd")
        else:
            check(lib.avdf_downsample_bwd(_p(dy), _p(dx), B, Tin, C, r, _stream()), "downsample_bwd")
        return dx, None, None


def upsample(x, r):
    return ResampleFn.apply(x, r, "up")


def downsample(x, r):
    return ResampleFn.apply(x, r, "down")


class AddUpsampledFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, small, r):
        a, small = _c(a), _c(small)
        B, T, C = a.shape
        out = torch.empty_like(a)
        check(lib.avdf_add_upsampled(_p(a), _p(small), _p(out), B, T, C, r, _stream()), "add_upsampled")
        ctx.cfg = (B, T, C, r)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, C, r = ctx.cfg
        dout = dout.contiguous()
        ds = torch.empty((B, T // r, C), dtype=dout.dtype, device=dout.device)
        check(lib.avdf_upsample_bwd(_p(dout), _p(ds), B, T // r, C, r, _stream()), "upsample_bwd")
        return dout, ds, None


def add_upsampled(a, small, r):
    return AddUpsampledFn.apply(a, small, r)


class ScaleRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        x, s = _c(x), _c(s)
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        check(lib.avdf_scale_rows(_p(x), _p(s), _p(y), M, C, _stream()), "scale_rows")
        ctx.save_for_backward(s)
        return y

    @staticmethod
    def backward(ctx, dy):
        (s,) = ctx.saved_tensors
        dy = dy.contiguous()
        C = dy.shape[-1]
        dx = torch.empty_like(dy)
        check(lib.avdf_scale_rows(_p(dy), _p(s), _p(dx), dy.numel() // C, C, _stream()), "scale_rows")
        return dx, None


def scale_rows(x, s):
    return ScaleRowsFn.apply(x, s)


def transpose_last2(x):
    """[B, R, C] -> [B, C, R] (no autograd: used at the plugin boundary only)."""
    x = _c(x)
    B, R, Cc = x.shape
    y = torch.empty((B, Cc, R), dtype=x.dtype, device=x.device)
    check(lib.avdf_transpose(_p(x), _p(y), B, R, Cc, _stream()), "transpose")
    return y


# --------------------------------------------------------------------------------------------
# pyramid glue: gapped level sequence for the shared heads, compaction of their outputs, mask pyramid
# --------------------------------------------------------------------------------------------
def _int_table(v):
    return (ctypes.c_int * len(v))(*[int(i) for i in v])


class GapConcatFn(torch.autograd.Function):
    """levels [B, T_l, C] x L (+ their masks [B, T_l] x L) -> ([B, sum(T_l + 1), C], gapped mask [B, sum(T_l + 1)]) with an
    all-zero row after every level (see meta_archs.gap_levels).  One launch forward, one backward."""

    @staticmethod
    def forward(ctx, n_levels, *args):
        xs = [_c(t) for t in args[:n_levels]]
        ms = [_c(t) for t in args[n_levels:]]
        assert len(ms) == n_levels
        B, _, C = xs[0].shape
        lens = [int(t.shape[1]) for t in xs]
        for t in xs + ms:
            _p(t)
        Tg = sum(lens) + n_levels
        out = torch.empty((B, Tg, C), dtype=torch.float32, device=xs[0].device)
        gmask = torch.empty((B, Tg), dtype=torch.float32, device=xs[0].device)
        check(lib.avdf_gap_concat_fwd(_ptr_table(xs), _ptr_table(ms), _int_table(lens), n_levels, _p(out), _p(gmask), B, C,
                                      _stream()), "gap_concat_fwd")
        ctx.cfg = (B, C, lens)
        ctx.mark_non_differentiable(gmask)
        ctx.set_materialize_grads(False)          # no zero-filled gradient tensor for the mask output
        return out, gmask

    @staticmethod
    def backward(ctx, dout, _dmask):
        B, C, lens = ctx.cfg
        dout = dout.contiguous()
        need = ctx.needs_input_grad[1:1 + len(lens)]
        dxs = [torch.empty((B, n, C), dtype=torch.float32, device=dout.device) if nd else None for n, nd in zip(lens, need)]
        check(lib.avdf_gap_concat_bwd(_p(dout), _ptr_table(dxs), _int_table(lens), len(lens), B, C, _stream()), "gap_concat_bwd")
        return (None, *dxs, *([None] * len(lens)))


def gap_concat(levels, masks):
    return GapConcatFn.apply(len(levels), *levels, *masks)


class FpnTailFn(torch.autograd.Function):
    """Everything of FPN1D behind the lateral convs, all levels at once (necks.py:62-93): top-down x2 add, masked depthwise
    k3, LayerNorm, into the gapped level sequence of the shared heads -- two launches forward, three backward
    (avdf_fpn_tail_fwd / _bwd).  Engine path / inference only: the parameter gradients are accumulated into the flat-buffer
    views (`main_grad`) with float atomics.  args: n, eps, then n laterals, n masks, n dw weights, n gammas, n betas."""

    @staticmethod
    def forward(ctx, n, eps, *args):
        lat = [_c(t) for t in args[0:n]]
        masks = [_c(t) for t in args[n:2 * n]]
        dws, gammas, betas = list(args[2 * n:3 * n]), list(args[3 * n:4 * n]), list(args[4 * n:5 * n])
        B, _, C = lat[0].shape
        lens = [int(t.shape[1]) for t in lat]
        Tg = sum(lens) + n
        dev = lat[0].device
        need = any(ctx.needs_input_grad)
        td = torch.empty((B, Tg, C), dtype=torch.float32, device=dev)
        y = torch.empty_like(td)
        gmask = torch.empty((B, Tg), dtype=torch.float32, device=dev)
        u = torch.empty_like(td) if need else None
        stats = torch.empty((2, B, Tg), dtype=torch.float32, device=dev) if need else None
        dwc, gc, bc = [_c(t) for t in dws], [_c(t).view(-1) for t in gammas], [_c(t).view(-1) for t in betas]
        for t in lat + masks + dwc + gc + bc:
            _p(t)
        ilens = _int_table(lens)
        check(lib.avdf_fpn_tail_fwd(n, ilens, _ptr_table(lat), _ptr_table(masks), _ptr_table(dwc), _ptr_table(gc),
                                    _ptr_table(bc), _p(td), _p(y), _p(gmask), _p(u), _p(stats[0]) if need else None,
                                    _p(stats[1]) if need else None, B, C, float(eps), _stream()), "fpn_tail_fwd")
        if need:
            ctx.save_for_backward(td, u, stats, *masks, *dwc, *gc)
            ctx.cfg = (n, lens, B, C)
            ctx.mains = ([_main_grad(t) for t in dws], [_main_grad(t) for t in gammas], [_main_grad(t) for t in betas])
        ctx.mark_non_differentiable(gmask)
        return y, gmask

    @staticmethod
    def backward(ctx, dy, _dmask):
        n, lens, B, C = ctx.cfg
        sv = ctx.saved_tensors
        td, u, stats = sv[0], sv[1], sv[2]
        masks, dwc, gc = list(sv[3:3 + n]), list(sv[3 + n:3 + 2 * n]), list(sv[3 + 2 * n:3 + 3 * n])
        mw, mg, mb = ctx.mains
        assert all(m is not None for m in mw + mg + mb), "FpnTailFn backward needs main_grad views on every FPN parameter"
        dy = dy.contiguous()
        dum, dtd = torch.empty_like(td), torch.empty_like(td)
        dlat = [torch.empty((B, lens[l], C), dtype=torch.float32, device=td.device) for l in range(n)]
        check(lib.avdf_fpn_tail_bwd(n, _int_table(lens), _p(dy), _p(u), _p(stats[0]), _p(stats[1]), _p(td), _ptr_table(masks),
                                    _ptr_table(dwc), _ptr_table(gc), _p(dum), _p(dtd), _ptr_table(dlat),
                                    _ptr_table([m.view(-1) for m in mw]), _ptr_table([m.view(-1) for m in mg]),
                                    _ptr_table([m.view(-1) for m in mb]), B, C, _stream()), "fpn_tail_bwd")
        return (None, None, *dlat, *([None] * (4 * n)))


def fpn_tail_fusable(laterals, dws, gammas, betas):
    """The fused FPN tail applies at inference and on the engine path (every FPN parameter accumulates into a flat-buffer
    view); levels must halve in length and be at most 256 channels wide."""
    C = laterals[0].shape[-1]
    lens = [int(t.shape[1]) for t in laterals]
    if not (FUSE_FPN and C <= 256 and C % 4 == 0 and len(laterals) <= 8 and all(lens[i] == 2 * lens[i + 1] for i in range(len(lens) - 1))):
        return False
    if not torch.is_grad_enabled():
        return True
    return all(_main_grad(t) is not None for t in list(dws) + list(gammas) + list(betas))


def fpn_tail(laterals, masks, dws, gammas, betas, eps=1e-5):
    n = len(laterals)
    return FpnTailFn.apply(n, eps, *laterals, *masks, *dws, *gammas, *betas)


FUSE_FPN = _os.environ.get("AVDF_FUSE_FPN", "1") != "0"


class HeadOutFn(torch.autograd.Function):
    """y [B, Tg, NC] on the gapped sequence -> [B, P, NC] in point order, optionally relu(y * scale_level)."""

    @staticmethod
    def forward(ctx, y, lens, relu, *scales):
        y = _c(y)
        B, Tg, NC = y.shape
        L = len(lens)
        sc = [s.contiguous().view(1) for s in scales] if scales else None
        out = torch.empty((B, Tg - L, NC), dtype=torch.float32, device=y.device)
        check(lib.avdf_head_out_fwd(_p(y), _ptr_table(sc) if sc else None, _int_table(lens), L, _p(out), B, NC, int(relu),
                                    _stream()), "head_out_fwd")
        ctx.save_for_backward(y, out, *(sc or []))
        ctx.cfg = (B, Tg, NC, list(lens), int(relu), [s.shape for s in scales])
        ctx.main = [_main_grad(s) for s in scales]
        return out

    @staticmethod
    def backward(ctx, dout):
        y, out, *sc = ctx.saved_tensors
        B, Tg, NC, lens, relu, sshapes = ctx.cfg
        L = len(lens)
        dout = dout.contiguous()
        dy = torch.empty_like(y)
        acc = bool(sc) and all(m is not None for m in ctx.main)
        ds = None
        if sc:
            ds = [m.view(1) for m in ctx.main] if acc else [torch.zeros(1, dtype=torch.float32, device=y.device) for _ in sc]
        check(lib.avdf_head_out_bwd(_p(dout), _p(y), _p(out), _ptr_table(sc) if sc else None, _ptr_table(ds) if ds else None,
                                    _int_table(lens), L, _p(dy), B, NC, relu, _stream()), "head_out_bwd")
        gs = [None] * len(sc) if (acc or not sc) else [d.view(shp) for d, shp in zip(ds, sshapes)]
        return (dy, None, None, *gs)


def head_out(y, lens, relu=False, scales=()):
    return HeadOutFn.apply(y, tuple(int(n) for n in lens), relu, *scales)


@torch.no_grad()
def droppath_keep(u, keep_prob):
    """u [n_modules, B] uniform draws, keep_prob [n_modules] -> floor(kp + u) / kp (blocks.py:1403-1409), one launch."""
    u = _c(u)
    out = torch.empty_like(u)
    check(lib.avdf_keep_from_uniform(_p(u), _p(_c(keep_prob).view(-1)), _p(out), u.numel(), int(u.shape[-1]), 0, 0.0, _stream()),
          "keep_from_uniform")
    return out


@torch.no_grad()
def dropout_keep(u, p):
    """u uniform draws -> (u >= p) / (1 - p): the keep mask of nn.Dropout(p), one launch."""
    u = _c(u)
    out = torch.empty_like(u)
    check(lib.avdf_keep_from_uniform(_p(u), None, _p(out), u.numel(), 1, 1, float(p), _stream()), "keep_from_uniform")
    return out


@torch.no_grad()
def mask_pyramid(mask, n_levels, stride=2):
    """[B, T] -> n_levels masks, level l = mask[:, ::stride^(l+1)] (one launch)."""
    mask = _c(mask)
    B, T = mask.shape
    outs, n = [], T
    for _ in range(n_levels):
        n //= stride
        outs.append(torch.empty((B, n), dtype=torch.float32, device=mask.device))
    check(lib.avdf_mask_pyramid(_p(mask), _ptr_table(outs), n_levels, B, T, stride, _stream()), "mask_pyramid")
    return outs


# --------------------------------------------------------------------------------------------
# InstanceNorm + LeakyReLU, pooling
# --------------------------------------------------------------------------------------------
class InstNormLReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, slope):
        x = _c(x)
        B, T, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((B, C), dtype=x.dtype, device=x.device)
        rstd = torch.empty((B, C), dtype=x.dtype, device=x.device)
        check(lib.avdf_instnorm_lrelu_fwd(_p(x), _p(y), _p(mean), _p(rstd), B, T, C, eps, slope, _stream()),
              "instnorm_lrelu_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        B, T, C = x.shape
        dx = torch.empty_like(x)
        check(lib.avdf_instnorm_lrelu_bwd(_p(dy.contiguous()), _p(x), _p(mean), _p(rstd), _p(dx), B, T, C, ctx.slope,
                                          _stream()), "instnorm_lrelu_bwd")
        return dx, None, None


def instnorm_lrelu(x, eps=1e-5, slope=0.2):
    return InstNormLReluFn.apply(x, eps, slope)


class MaxAvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, T, C = x.shape
        y = torch.empty((B, 2 * C), dtype=x.dtype, device=x.device)
        check(lib.avdf_maxavgpool_fwd(_p(x), _p(y), B, T, C, _stream()), "maxavgpool_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, T, C = x.shape
        dx = torch.empty_like(x)
        check(lib.avdf_maxavgpool_bwd(_p(dy.contiguous()), _p(x), _p(dx), B, T, C, _stream()), "maxavgpool_bwd")
        return dx


def maxavgpool(x):
    return MaxAvgPoolFn.apply(x)


# --------------------------------------------------------------------------------------------
# narrow-output conv
# --------------------------------------------------------------------------------------------
class SmallConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, mask, scale):
        ctx.main = (_main_grad(w), _main_grad(bias), _main_grad(scale))
        x, w, bias, mask = _c(x), _c(w), _c(bias), _c(mask)
        B, T, Cin = x.shape
        Cout, _, K = w.shape
        sc = None if scale is None else scale.contiguous().view(1)
        y = torch.empty((B, T, Cout), dtype=x.dtype, device=x.device)
        check(lib.avdf_smallconv_fwd(_p(x), _p(w), _p(bias), _p(mask), _p(sc), _p(y), B, T, Cin, Cout, K, _stream()),
              "smallconv_fwd")
        ctx.save_for_backward(x, w, mask, sc, y)
        ctx.cfg = (B, T, Cin, Cout, K, bias is not None, None if scale is None else scale.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mask, sc, y = ctx.saved_tensors
        B, T, Cin, Cout, K, has_bias, sshape = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        mw, mb, ms = ctx.main
        acc = 1 if (mw is not None and (not has_bias or mb is not None) and (sc is None or ms is not None)) else 0
        dw = mw.view(w.shape) if acc else torch.empty_like(w)
        db = ds = None
        if has_bias:
            db = mb.view(Cout) if acc else torch.empty(Cout, dtype=w.dtype, device=w.device)
        if sc is not None:
            ds = ms.view(1) if acc else torch.empty(1, dtype=w.dtype, device=w.device)
        ws = _ws(lib.avdf_smallconv_bwd_ws(B, T, Cin, Cout, K), x)
        check(lib.avdf_smallconv_bwd(_p(dy), _p(x), _p(w), _p(y), _p(mask), _p(sc), _p(dx), _p(dw), _p(db), _p(ds),
                                     _p(ws), B, T, Cin, Cout, K, acc, _stream()), "smallconv_bwd")
        if acc:
            return dx, None, None, None, None
        return dx, dw, db, None, None if ds is None else ds.view(sshape)


def smallconv(x, w, bias=None, mask=None, scale=None):
    return SmallConvFn.apply(x, w, bias, mask, scale)


# --------------------------------------------------------------------------------------------
# K1: interpolation + concat (no autograd: inputs are frozen features)
# --------------------------------------------------------------------------------------------
def interp_concat(streams, out_len, out=None):
    """streams: list of [T_s, C_s] time-major fp32 device tensors -> [out_len, sum C_s]."""
    ctot = sum(int(s.shape[1]) for s in streams)
    if out is None:
        out = torch.empty((out_len, ctot), dtype=torch.float32, device=streams[0].device)
    assert out.shape == (out_len, ctot) and out.is_contiguous()
    off = 0
    for s in streams:
        s = _c(s)
        check(lib.avdf_interp_linear(_p(s), _p(out), int(s.shape[0]), int(s.shape[1]), out_len, ctot, off, _stream()),
              "interp_linear")
        off += int(s.shape[1])
    return out


def interp_concat_batched(stream_bufs, out):
    """stream_bufs: list of [B, T_s, C_s] tensors (one per stream kind); out [B, T, sum C_s]: one launch per stream."""
    B, T, ctot = out.shape
    assert out.is_contiguous()
    off = 0
    for s in stream_bufs:
        assert s.is_contiguous() and s.shape[0] == B
        Ts, Cs = int(s.shape[1]), int(s.shape[2])
        check(lib.avdf_interp_linear_batched(_p(s), _p(out), B, Ts * Cs, T * ctot, Ts, Cs, T, ctot, off, _stream()),
              "interp_linear_batched")
        off += Cs
    return out


# --------------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------------
class LocLossFn(torch.autograd.Function):
    """Returns out[5] = (cls, reg, video, final, num_pos); only out[3] carries gradient."""

    @staticmethod
    def forward(ctx, logits, offsets, video_logits, points, pmask, gt, ngt, state, cfg, num_pos_hook):
        logits, offsets, video_logits = _c(logits), _c(offsets), _c(video_logits)
        B, P = logits.shape
        G = gt.shape[1]
        radius, center_sample, ls, lw, vw, mom = cfg
        st = _stream()
        ws = _ws(lib.avdf_loc_losses_ws(B, P), logits)
        num_pos = torch.empty(1, dtype=torch.float32, device=logits.device)
        check(lib.avdf_loc_assign(_p(points), _p(pmask), _p(gt), _p(ngt), _p(ws), _p(num_pos), B, P, G, radius,
                                  int(center_sample), st), "loc_assign")
        if num_pos_hook is not None:
            num_pos_hook(num_pos)              # data parallel: all-reduce(sum) of the positive count
        out = torch.empty(5, dtype=torch.float32, device=logits.device)
        dlog = torch.empty_like(logits)
        doff = torch.empty_like(offsets)
        dvid = torch.empty_like(video_logits)
        check(lib.avdf_loc_losses(_p(logits), _p(offsets), _p(pmask), _p(ngt), _p(video_logits), _p(ws), _p(num_pos),
                                  _p(state), _p(out), _p(dlog), _p(doff), _p(dvid), _p(ws[B * P * 3:]), B, P,
                                  ls, lw, vw, mom, st), "loc_losses")
        ctx.save_for_backward(dlog, doff, dvid)
        return out

    @staticmethod
    def backward(ctx, dout):
        dlog, doff, dvid = ctx.saved_tensors
        g = dout[3]
        return dlog * g, doff * g, dvid * g, None, None, None, None, None, None, None


def loc_losses(logits, offsets, video_logits, points, pmask, gt, ngt, state, cfg, num_pos_hook=None):
    return LocLossFn.apply(logits, offsets, video_logits, points, pmask, gt, ngt, state, cfg, num_pos_hook)


class FocalSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        x, t = _c(x), _c(t)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(lib.avdf_focal_loss_sum(_p(x), _p(t), _p(out), _p(dx), x.numel(), _stream()), "focal_loss_sum")
        ctx.save_for_backward(dx)
        return out

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None


def focal_loss_sum(x, t):
    return FocalSumFn.apply(x, t)


# --------------------------------------------------------------------------------------------
# decode + NMS
# --------------------------------------------------------------------------------------------
def decode_nms(points, logits, offsets, pmask, pre_nms_thresh, duration_thresh, min_score, iou_thr, voting_thr,
               max_seg):
    B, P = logits.shape
    dev = logits.device
    segs = torch.empty((B, max_seg, 2), dtype=torch.float32, device=dev)
    scores = torch.empty((B, max_seg), dtype=torch.float32, device=dev)
    pidx = torch.empty((B, max_seg), dtype=torch.int32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    check(lib.avdf_decode_nms(_p(points), _p(_c(logits)), _p(_c(offsets)), _p(_c(pmask)), _p(segs), _p(scores),
                              _p(pidx), _p(count), B, P, pre_nms_thresh, duration_thresh, min_score, iou_thr,
                              voting_thr, max_seg, _stream()), "decode_nms")
    return segs, scores, pidx, count


# package default: exact fp32 everywhere; AVDF_GEMM_MODE=1 switches every forward GEMM of the process to bf16x3
# (the library reads the variable itself for threads that never call set_gemm_mode)
