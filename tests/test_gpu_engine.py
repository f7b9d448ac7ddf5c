"""engine.TrainStep / InferStep (flat buffers, hipGraphs, fused optimiser) against the plain `model(video_list)` path."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(video_input_dim=8, audio_input_dim=88, max_seq_len=192, embd_dim=32, fpn_dim=32, head_dim=32, n_head=4)
STREAMS = [(192, 8), (96, 64), (383, 24)]


def _model(seed=0):
    from audio_visual_deepfake_detection_amd import build_model
    torch.manual_seed(seed)
    model, cfg = build_model("deepfake_exp10.yaml", **SMALL)
    model = model.to("cuda:0")
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    model.interpolator.dp1.p = 0.0
    return model, cfg


def _fill(eng, seed, all_real=False):
    g = torch.Generator().manual_seed(seed)
    clips = []
    for b in range(eng.B):
        ss = [torch.randn(shp, generator=g) for shp in STREAMS]
        for dst, src in zip(eng.streams[b], ss):
            dst.copy_(src)
        segs = None
        if b % 2 == 0 and not all_real:
            segs = torch.tensor([[20.0 + 7 * b, 60.0 + 5 * b], [100.0, 118.0]])
        if hasattr(eng, "gt"):
            eng.gt[b].zero_()
            eng.ngt[b] = 0 if segs is None else len(segs)
            if segs is not None:
                eng.gt[b, :len(segs)].copy_(segs)
        clips.append(dict(video_id=f"c{b}", streams=[s.numpy() for s in ss], crop=None, segments=segs,
                          labels=None if segs is None else torch.zeros(len(segs), dtype=torch.int64), fps=25.0,
                          duration=192 / 25.0, feat_stride=1.0, feat_num_frames=1.0))
    return clips


@pytest.mark.parametrize("use_graph", [False, True])
def test_train_step_equals_plain_model_path(use_graph):
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    model, cfg = _model(1)
    ref = copy.deepcopy(model).train()
    eng = TrainStep(model, 4, STREAMS, lr=1e-3, weight_decay=0.05, clip_grad_l2norm=1.0, use_graph=use_graph)
    clips = _fill(eng, 5)
    eng.capture()
    eng.step()
    got = eng.losses()
    losses = ref(clips)
    losses["final_loss"].backward()
    for k in ("cls_loss", "reg_loss", "reco_cls_loss", "final_loss"):
        ref_k = float(losses[k].detach())
        assert abs(got[k] - ref_k) < 2e-5 * max(1.0, abs(ref_k)), (k, got[k], ref_k)
    # one fused step == clip_grad_norm_ + torch.optim.AdamW (decay / no-decay groups of make_optimizer) + EMA 0.999
    from audio_visual_deepfake_detection_amd.engine import split_decay
    named = dict(ref.named_parameters())
    live = [n for n, p in named.items() if p.grad is not None]
    decay, no_decay = split_decay(ref)
    torch.nn.utils.clip_grad_norm_([named[n] for n in live], 1.0)
    opt = torch.optim.AdamW([{"params": [named[n] for n in decay if n in live], "weight_decay": 0.05},
                             {"params": [named[n] for n in no_decay if n in live], "weight_decay": 0.0}], lr=1e-3)
    before = {n: named[n].detach().clone() for n in live}
    opt.step()
    new = dict(model.named_parameters())
    ema = eng.ema_state_dict()
    worst = 0.0
    for n in live:
        upd_ref, upd_got = named[n].detach() - before[n], new[n].detach() - before[n]
        scale = float(upd_ref.abs().max()) + 1e-12
        if float(named[n].grad.abs().max()) < 1e-7:
            continue        # mathematically zero gradients (key bias under softmax, conv bias under InstanceNorm): Adam
                            # turns their rounding noise into +-lr steps in either implementation
        worst = max(worst, float((upd_got - upd_ref).abs().max()) / scale)
        e_ref = 0.999 * before[n] + 0.001 * named[n].detach()
        assert float((ema[n] - e_ref).abs().max()) <= 1e-6 * (1.0 + float(e_ref.abs().max())), n
    assert worst < 2e-2, worst            # Adam's first step is ~lr*sign(g): elements with |g| near eps differ visibly


def test_all_real_batch_has_no_positive_points_and_stays_finite():
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    model, _ = _model(2)
    eng = TrainStep(model, 2, STREAMS, use_graph=True)
    _fill(eng, 6, all_real=True)
    eng.capture()
    for _ in range(2):
        eng.step()
    out = eng.losses()
    assert out["num_pos"] == 0.0 and out["reg_loss"] == 0.0
    assert all(np.isfinite(v) for v in out.values())
    assert torch.isfinite(eng.flat_p).all()


def test_infer_step_equals_eval_model():
    from audio_visual_deepfake_detection_amd.engine import InferStep
    model, _ = _model(3)
    model.allow_batched_eval = True
    eng = InferStep(model, 3, STREAMS, use_graph=True)
    clips = _fill(eng, 7)
    eng.capture()
    segs, scores, pidx, count, cls_scores = eng.step()
    for b, clip in enumerate(clips):
        res = model.eval()([clip])[0]
        n = int(count[b])
        assert n == res["segments"].shape[0]
        assert sorted(pidx[b, :n].tolist()) == sorted(res["kept_points"].tolist())
        assert torch.allclose(scores[b, :n].cpu().sort().values, res["scores"].sort().values, atol=1e-5)


def test_audio_only_workload_steps():
    """bench.py --workload audio (BASELINE config 2 shapes): one feature stream, video_input_dim = 0 -- the engine takes a
    single-stream K1 and the step stays finite and changes the weights."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    torch.manual_seed(3)
    small = dict(SMALL, video_input_dim=0, audio_input_dim=64)
    model, cfg = build_model("deepfake_exp10.yaml", **small)
    model = model.to("cuda:0")
    eng = TrainStep(model, 2, [(96, 64)], lr=1e-3, use_graph=True)
    g = torch.Generator().manual_seed(5)
    for b in range(2):
        eng.streams[b][0].copy_(torch.randn(96, 64, generator=g))
        eng.gt[b].zero_()
        eng.ngt[b] = 0
    eng.gt[0, 0] = torch.tensor([30.0, 80.0])
    eng.ngt[0] = 1
    eng.capture()
    before = eng.flat_p.clone()
    for _ in range(2):
        eng.step()
    torch.cuda.synchronize()
    ls = eng.losses()
    assert all(np.isfinite(v) for v in ls.values()), ls
    assert ls["num_pos"] > 0
    assert float((eng.flat_p - before).abs().max()) > 0


def test_ten_step_trajectory_matches_oracle_with_torch_adamw():
    """Ten engine steps queued back to back (no host synchronisation between them: the lr / bias-correction scalars of every
    step travel through their own pinned slot) under a warm-up + cosine schedule, against the CPU oracle driven by
    clip_grad_norm_ + torch.optim.AdamW (the reference's step, train_utils.py:302-316): loss trajectory, loss normaliser and
    final weights.  deterministic=True: slab reducers instead of float atomics -> two runs are bit-identical."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.engine import TrainStep, split_decay
    from audio_visual_deepfake_detection_amd.libs.utils.lr_schedulers import LinearWarmupCosineAnnealingLR
    from oracle import ummaformer_ref as R
    cfg = R.make_cfg(input_dim=96, max_seq_len=192, embd_dim=32, n_head=4, fpn_dim=32, head_dim=32)
    sd0 = R.synth_state_dict(R.param_shapes(cfg), 29, True)
    sched = LinearWarmupCosineAnnealingLR(2e-3, 3, 12)
    nstep, B = 10, 4

    def run_engine():
        model, _ = build_model("deepfake_exp10.yaml", **SMALL)
        model.load_state_dict(sd0)
        model = model.to("cuda:0")
        for mod in model.modules():
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        model.interpolator.dp1.p = 0.0
        eng = TrainStep(model, B, STREAMS, lr=0.0, weight_decay=0.05, clip_grad_l2norm=1.0, use_graph=True, deterministic=True)
        eng.set_lr_schedule(sched)
        clips = _fill(eng, 41)
        eng.capture()
        traj = []
        for _ in range(nstep):
            eng.step()
            traj.append(eng.loss_out.clone())              # device-side copy: no host sync inside the loop
        torch.cuda.synchronize()
        return eng, model, clips, torch.stack(traj).cpu()

    eng, model, clips, traj = run_engine()
    eng2, _, _, traj2 = run_engine()
    assert torch.equal(traj, traj2) and torch.equal(eng.flat_p, eng2.flat_p), "deterministic engine runs differ"

    # --- oracle + torch.optim.AdamW on the CPU ---------------------------------------------------------------
    torch.set_num_threads(8)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
    decay, no_decay = split_decay(model)
    dead = set(eng.dead)
    opt = torch.optim.AdamW([{"params": [sd[n] for n in decay if n not in dead], "weight_decay": 0.05},
                             {"params": [sd[n] for n in no_decay if n not in dead], "weight_decay": 0.0}], lr=1e-3)
    x = torch.stack([R.interp_concat(c["streams"], 192) for c in clips])
    mask = torch.ones((B, 1, 192))
    gts = [c["segments"] for c in clips]
    gls = [c["labels"] for c in clips]
    norm = cfg["init_loss_norm"]
    live = [sd[n] for n in sd if n not in dead]
    for it in range(nstep):
        for g in opt.param_groups:
            g["lr"] = sched(it)
        opt.zero_grad(set_to_none=True)
        out = R.model_forward(sd, cfg, x, mask, training=True)
        ls, norm, npos = R.losses(cfg, out, gts, gls, norm)
        ls["final_loss"].backward()
        torch.nn.utils.clip_grad_norm_(live, 1.0)
        opt.step()
        got = traj[it]
        # two fp32 implementations drift apart along an Adam trajectory (every element's step is normalised to ~lr, so
        # rounding noise in small gradients is amplified step by step): tight over the first steps, looser afterwards
        tol = 2e-4 if it < 5 else 3e-3
        for j, k in enumerate(("cls_loss", "reg_loss", "reco_cls_loss", "final_loss")):
            ref_k = float(ls[k].detach())
            assert abs(float(got[j]) - ref_k) < tol * max(1.0, abs(ref_k)), (it, k, float(got[j]), ref_k)
        assert float(got[4]) == float(npos)
    assert abs(model.loss_normalizer - norm) < 1e-3
    # weights after ten steps: the total update of all live parameters points the same way and has the same size.  (Element by
    # element Adam is chaotic: every element's step is normalised to ~lr, so the rounding noise of small gradients becomes
    # +-lr steps in either implementation; the loss trajectory above is the functional check.)
    new = dict(model.named_parameters())
    d_ref = torch.cat([(sd[n].detach() - sd0[n]).flatten() for n in sd if n not in dead]).double()
    d_got = torch.cat([(new[n].detach().cpu() - sd0[n]).flatten() for n in sd if n not in dead]).double()
    cos = float((d_ref * d_got).sum() / (d_ref.norm() * d_got.norm()))
    assert cos > 0.98 and abs(float(d_got.norm() / d_ref.norm()) - 1.0) < 0.02, (cos, float(d_got.norm() / d_ref.norm()))


def test_engine_step_of_exp5_includes_the_reconstruction_loss():
    """exp5 (`AVLocPointTransformerRecoveryNoNorm`): the L1 reconstruction term on the real clips (av_fd_meta_arch.py:683-701)
    is part of final_loss and sends gradient into interpolator.expansion.* -- the engine step must equal the plain
    `model(video_list)` path, and the Expansion parameters must be live."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    torch.manual_seed(4)
    model, _ = build_model("deepfake_exp5.yaml", **SMALL)
    model = model.to("cuda:0")
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    model.interpolator.dp1.p = 0.0
    ref = copy.deepcopy(model).train()
    eng = TrainStep(model, 4, STREAMS, lr=1e-3, use_graph=True)
    assert eng.has_reco and not any(n.startswith("interpolator.expansion") for n in eng.dead)
    assert [b[0] for b in eng.bucket_ranges] == ["post", "pre"]
    clips = _fill(eng, 8)
    eng.capture()
    eng.step()
    got = eng.losses()
    losses = ref(clips)
    for k in ("cls_loss", "reg_loss", "reco_cls_loss", "final_loss", "reco_loss"):
        ref_k = float(losses[k].detach()) if torch.is_tensor(losses[k]) else float(losses[k])
        assert abs(got[k] - ref_k) < 2e-5 * max(1.0, abs(ref_k)), (k, got[k], ref_k)
    losses["final_loss"].backward()
    # Adam's first moment after one step = 0.1 * clipped gradient: linear in the gradient, comparable element-wise
    named = dict(ref.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    cs = min(1.0, 1.0 / (gn + 1e-6))
    for n in ("interpolator.expansion.up_1.conv_transpose.conv.weight", "interpolator.expansion.up_5.conv_transpose.conv.bias",
              "backbone.resselfattention.attn.key.weight", "interpolator.contraction.down_2.conv_block.conv.weight"):
        off, k = eng._offsets[n], named[n].numel()
        m1 = eng.exp_avg[off:off + k].view(named[n].shape)
        want = 0.1 * cs * named[n].grad
        assert float((m1 - want).abs().max()) < 1e-3 * float(want.abs().max()) + 1e-9, n


def test_probe_leaves_inputs_and_rng_untouched():
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    model, _ = _model(6)
    torch.manual_seed(77)
    a = torch.rand(3, device="cuda:0")
    torch.manual_seed(77)
    eng = TrainStep(model, 2, STREAMS, use_graph=False)
    b = torch.rand(3, device="cuda:0")
    assert torch.equal(a, b), "constructing the engine consumed device RNG"
    assert float(eng.gt.abs().sum()) == 0.0 and int(eng.ngt.sum()) == 0 and float(eng.x.abs().sum()) == 0.0


def _engine_gradients(fwd, bwd, B=2):
    """Gradients of one full-size exp10 step (eager stages without the optimiser pass) as {name: tensor}, plus losses."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.engine import TrainStep
    shapes = [(768, 256), (383, 2048), (1535, 768)]
    torch.manual_seed(3)
    model, _ = build_model("deepfake_exp10.yaml")
    model = model.to("cuda:0")
    for mod in model.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    model.interpolator.dp1.p = 0.0
    eng = TrainStep(model, B, shapes, lr=1e-3, use_graph=False, fwd_bf16x3=fwd, bwd_bf16x3=bwd)
    g = torch.Generator().manual_seed(11)
    for b in range(B):
        for dst in eng.streams[b]:
            dst.copy_(torch.randn(dst.shape, generator=g))
        eng.gt[b].zero_()
        eng.ngt[b] = 0
    eng.gt[0, 0, 0], eng.gt[0, 0, 1], eng.ngt[0] = 100.0, 260.0, 1
    with torch.cuda.stream(eng.stream):
        eng._stage_a()
        eng._stage_loss()
        for i in range(eng.n_phases):
            eng._stage_bwd(i)
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    grads = {n: eng.flat_g[off:off + named[n].numel()].detach().cpu().double().clone() for n, off in eng._offsets.items()}
    return grads, eng.losses()


def test_bf16x3_training_mode_gradient_tolerance():
    """The opt-in split-bf16 training mode (forward AND backward GEMMs: data gradients of every conv, weight gradients of the
    K = 1 layers) against the exact-fp32 engine on a full-size exp10 step.  Stated tolerance: losses to 1e-5 relative, the
    global gradient norm to 1e-4 relative, every parameter tensor's gradient to 2e-3 of its L2 norm (single elements can
    move more: a LeakyReLU / ReLU derivative flips when a pre-activation crosses zero at rounding level)."""
    g0, l0 = _engine_gradients(False, False)
    g1, l1 = _engine_gradients(True, True)
    for k in ("cls_loss", "reg_loss", "final_loss"):
        assert abs(l0[k] - l1[k]) <= 1e-5 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    n0 = sum(float((v ** 2).sum()) for v in g0.values()) ** 0.5
    n1 = sum(float((v ** 2).sum()) for v in g1.values()) ** 0.5
    assert abs(n0 - n1) <= 1e-4 * n0, (n0, n1)
    rels = []
    for n, a in g0.items():
        na = float(a.norm())
        if na < 1e-5 * n0 / len(g0) ** 0.5 or _zero_by_construction(n):
            continue                                  # mathematically zero gradients (rounding noise in either mode)
        rels.append((float((g1[n] - a).norm()) / na, n))
    rels.sort(reverse=True)
    assert sum(r > 0 for r, _ in rels) > 100, "the bf16x3 kernels were not selected"
    assert rels[0][0] <= 2e-3, rels[:12]


def _zero_by_construction(name):
    # a conv bias in front of an InstanceNorm over time and the key bias under the softmax have zero gradient
    # mathematically; what the kernels produce there is cancellation noise in either arithmetic
    return name.endswith("conv_block.conv.bias") or name.endswith(".key.bias")


@pytest.mark.parametrize("use_graph", [False, True])
def test_state_dict_resume_is_bit_exact_and_loads_into_torch_adamw(use_graph):
    """TrainStep.state_dict() / load_state_dict() (reference: save_checkpoint, libs/utils/train_utils.py:43-56; resume,
    train.py:115-133): 3 steps -> save -> FRESH engine (other initial weights) -> load -> 3 steps equals 6 straight
    steps bit for bit in deterministic mode (drop-path active: its RNG stream is part of the state), and the
    `optimizer` entry loads into a torch.optim.AdamW built with make_optimizer's partition."""
    from audio_visual_deepfake_detection_amd import build_model
    from audio_visual_deepfake_detection_amd.engine import TrainStep, split_decay

    def engine(seed):
        torch.manual_seed(seed)
        model, _ = build_model("deepfake_exp10.yaml", **SMALL)
        eng = TrainStep(model.to("cuda:0"), 4, STREAMS, lr=1e-3, weight_decay=0.05, use_graph=use_graph, deterministic=True)
        _fill(eng, 11)
        eng.capture()
        return eng

    torch.cuda.manual_seed(77)
    a = engine(1)
    for _ in range(3):
        a.step()
    ckpt = a.state_dict()
    mid_losses = a.losses()
    for _ in range(3):
        a.step()
    torch.cuda.synchronize()
    want_p, want_m, want_v, want_e = a.flat_p.clone(), a.exp_avg.clone(), a.exp_avg_sq.clone(), a.ema.clone()
    want_norm, want_losses = a.model._loss_norm_state.clone(), a.losses()

    # the checkpoint has the reference's keys and torch.optim's layout
    assert {"state_dict", "state_dict_ema", "optimizer"} <= set(ckpt)
    decay, no_decay = split_decay(a.model)
    assert [len(g["params"]) for g in ckpt["optimizer"]["param_groups"]] == [len(decay), len(no_decay)]
    assert all(float(s["step"]) == 3.0 for s in ckpt["optimizer"]["state"].values())
    assert len(ckpt["optimizer"]["state"]) == len(a.live_names)

    b = engine(2)                                   # different initial weights, different RNG position
    assert not torch.equal(b.flat_p, a.flat_p)
    b.load_state_dict(ckpt)
    assert b.step_count == 3 and b.losses is not None
    for _ in range(3):
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(b.flat_p, want_p) and torch.equal(b.exp_avg, want_m) and torch.equal(b.exp_avg_sq, want_v)
    assert torch.equal(b.ema, want_e) and torch.equal(b.model._loss_norm_state, want_norm)
    assert b.losses() == want_losses and b.losses() != mid_losses

    # hand-over to torch.optim.AdamW built by the reference's partition rules (train_utils.py:122-134)
    ref, _ = build_model("deepfake_exp10.yaml", **SMALL)
    ref = ref.to("cuda:0")
    ref.load_state_dict(ckpt["state_dict"])
    named = dict(ref.named_parameters())
    opt = torch.optim.AdamW([{"params": [named[n] for n in sorted(decay)], "weight_decay": 0.05},
                             {"params": [named[n] for n in sorted(no_decay)], "weight_decay": 0.0}], lr=1e-3)
    opt.load_state_dict(ckpt["optimizer"])
    n0 = a.live_names[0]
    st = opt.state[named[n0]]
    off, k = a._offsets[n0], named[n0].numel()
    assert float(st["step"]) == 3.0 and st["exp_avg"].shape == named[n0].shape
    # and back: the torch optimiser's own state_dict resumes the engine to the same trajectory
    c = engine(3)
    c.load_state_dict(dict(ckpt, optimizer=opt.state_dict(), state_dict={"module." + k_: v for k_, v in ckpt["state_dict"].items()}))
    for _ in range(3):
        c.step()
    torch.cuda.synchronize()
    assert torch.equal(c.flat_p, want_p) and torch.equal(c.exp_avg_sq, want_v)


def test_package_imported_before_torch_still_launches_kernels():
    """`import audio_visual_deepfake_detection_amd` as the FIRST import of a process (no torch yet -- what
    __graft_entry__.build() does before smoke()): the library must end up on the HIP runtime torch uses.  Without the
    import-order guard in _lib.py the first launch failed with "no ROCm-capable device is detected"."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import audio_visual_deepfake_detection_amd as P\n"
            "from audio_visual_deepfake_detection_amd import ops\n"
            "import torch\n"
            "x = torch.arange(2 * 8 * 16, dtype=torch.float32, device='cuda:0').view(2, 8, 16)\n"
            "y = ops.transpose_last2(x)\n"
            "assert torch.equal(y, x.transpose(1, 2).contiguous())\n"
            "print('ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
