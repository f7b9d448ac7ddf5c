"""Host runtime for one training step of the UMMAFormer path on one MI355X (one process per GPU).

The reference's step is `losses = model(video_list); losses['final_loss'].backward();
clip_grad_norm_; optimizer.step(); scheduler.step(); model_ema.update()` (libs/utils/train_utils.py:
302-316) driven from Python with one aten launch per op.  Here the same step is a short sequence of
hipGraphs over hand-written kernels, replayed on the caller's HIP stream (a stream hop per step costs
0.35 ms on this stack), with the data-parallel exchanges between them:

    graph A   : K1 interpolation+concat of the raw feature streams -> network forward -> label assignment
    (DP)      : all-reduce(sum) of [#positive points, #real clips]              [2 floats, RCCL]
    graph B0  : fused losses (+ their gradients); backward of the video-level front (DeepInterpolator)
    (DP)      :   all-reduce(sum) of gradient bucket 0, asynchronous ------------------+
    graph B1  : backward of heads, FPN and every block after the stem output          | overlapped
    (DP)      :   all-reduce(sum) of gradient bucket 1, asynchronous ---------------+  |
    graph B2  : backward of stem, cross-attention block and embedding              |  |
    (DP)      :   all-reduce(sum) of gradient bucket 2; wait for all three <--------+--+
    graph C   : grad-norm -> clipped AdamW -> weight EMA -> gradient reset, one pass over flat buffers

The backward pass is cut at the stem output (the one tensor every later block depends on): the network
is evaluated as two autograd graphs joined by a detached copy, so each phase is an ordinary
`torch.autograd.backward` and the flat gradient buffer is laid out bucket by bucket in the order the
phases complete.  At world size 1 there are no collectives and the whole sequence is captured as ONE graph.

Semantics kept from the reference: the focal / DIoU sums are divided by the EMA of the GLOBAL number
of positives (av_fd_no_recon.py:645-648,662-667), so gradients are SUM-reduced across ranks, not
averaged; exp5's L1 reconstruction loss is a mean over the real clips of the GLOBAL batch
(av_fd_meta_arch.py:683-701); parameters that never receive a gradient (interpolator.expansion.* in
exp10, backbone.hh_branch.<last>) are skipped by the optimiser exactly like torch.optim skips
`grad is None`; weight decay applies to conv/linear weights only (train_utils.py:73-111).
"""
import ctypes
import json
import os

import torch

from . import dp, ops
from ._lib import lib, check


def split_decay(model):
    """(decay, no_decay) parameter names with make_optimizer's rules (train_utils.py:73-111); pinned to the reference's
    own partition of its three shipped models by tests/golden/optimizer_split.npz."""
    from .libs.modeling.blocks import LayerNorm
    decay, no_decay = [], []
    for mn, m in model.named_modules():
        for pn, p in m.named_parameters(recurse=False):
            fpn = f"{mn}.{pn}" if mn else pn
            if pn.endswith("bias") or pn.endswith("scale") or pn.endswith("rel_pe") or pn.endswith("time_weighting"):
                no_decay.append(fpn)
            elif pn.endswith("weight") and isinstance(m, (LayerNorm, torch.nn.GroupNorm, torch.nn.Embedding,
                                                          torch.nn.LayerNorm)):
                no_decay.append(fpn)
            elif pn.endswith("weight"):
                decay.append(fpn)
            else:
                raise AssertionError(f"parameter {fpn} was not separated into either decay/no_decay set")
    return decay, no_decay


def _pad4(n):
    return (n + 3) // 4 * 4


class TrainStep:
    """Owns the flat parameter / gradient / optimiser-state buffers and the captured step.

    deterministic=True : parameter gradients are produced by the slab reducers (no float atomics) and summed by
                         autograd into the flat buffer -- bit-reproducible steps, a few % slower.
    overlap=True       : gradient buckets are all-reduced asynchronously while the next backward phase runs.
    """

    HP_SLOTS = 4

    def __init__(self, model, batch, streams_shapes, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8,
                 clip_grad_l2norm=1.0, ema_decay=0.999, use_graph=True, process_group=None, max_gt=8, wgrad_side=False,
                 fwd_bf16x3=False, bwd_bf16x3=False, deterministic=False, overlap=True):
        assert next(model.parameters()).is_cuda
        self.model = model.train()
        self.dev = next(model.parameters()).device
        self.B, self.T, self.C = batch, model.max_seq_len, model.input_dim
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.clip, self.ema_decay = clip_grad_l2norm, ema_decay
        self.pg = process_group
        self.world = 1 if process_group is None else torch.distributed.get_world_size(process_group)
        self.use_graph = use_graph
        self.wgrad_side = wgrad_side        # weight-gradient GEMMs on a second stream (joined at the end of backward)
        self.fwd_bf16x3 = fwd_bf16x3        # forward GEMMs with split-bf16 operands (faster; gradients ~1e-3 instead of 1e-5)
        self.bwd_bf16x3 = bwd_bf16x3        # backward GEMMs (dgrad, K = 1 wgrad) with split-bf16 operands (opt-in as well)
        self.deterministic = deterministic
        self.overlap = overlap
        self.step_count = 0
        self.comm_events = None             # bench.py: (backward done, all-reduces done) events of the last step
        dev = self.dev
        # the probe, the warm-up, the capture and EAGER steps all run on this stream, so autograd's per-parameter
        # accumulation nodes never straddle two streams (a node created during capture and reused by an eager step on
        # another stream makes autograd warn about -- and insert -- cross-stream synchronisation).  Captured graphs are
        # replayed on the caller's stream: a stream hop per step costs ~0.35 ms on this stack (measured: 686 vs 707 clips/s)
        self.stream = torch.cuda.Stream(device=dev)

        # ---- static inputs (raw per-clip feature streams live in HBM, time-major [T_s, C_s]) ----------
        # one [B, T_s, C_s] buffer per stream kind (K1 runs as one launch per kind); streams[b][s] are views
        self.stream_bufs = [torch.zeros((batch,) + tuple(shp), dtype=torch.float32, device=dev) for shp in streams_shapes]
        self.streams = [[buf[b] for buf in self.stream_bufs] for b in range(batch)]
        assert sum(s[1] for s in streams_shapes) == self.C
        self.x = torch.zeros((batch, self.T, self.C), dtype=torch.float32, device=dev)
        self.mask = torch.ones((batch, self.T), dtype=torch.float32, device=dev)
        self.gt = torch.zeros((batch, max_gt, 2), dtype=torch.float32, device=dev)
        self.ngt = torch.zeros((batch,), dtype=torch.int32, device=dev)
        lens = [self.T // s for s in model.fpn_strides]
        self.points = torch.cat(model.point_generator(lens), dim=0).contiguous()
        self.P = int(self.points.shape[0])

        if self.world > 1:
            dp.broadcast_params_(model, src=0, group=self.pg)       # one replica's initial weights on every rank

        self.counts = torch.zeros(2, dtype=torch.float32, device=dev)      # [#positive points, #real clips] (global)
        self.loss_out = torch.zeros(6, dtype=torch.float32, device=dev)    # cls, reg, video, final, num_pos, reco
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hp = torch.zeros(3, dtype=torch.float32, device=dev)
        self._hp_host = [torch.zeros(3, dtype=torch.float32).pin_memory() for _ in range(self.HP_SLOTS)]
        self._hp_done = [None] * self.HP_SLOTS
        self.ws_loss = torch.empty(int(lib.avdf_loc_losses_ws(batch, self.P)), dtype=torch.float32, device=dev)
        self.ws_sumsq = torch.empty(int(lib.avdf_sumsq_ws()), dtype=torch.float32, device=dev)
        self.ws_l1 = torch.empty(int(lib.avdf_l1_mean_ws()), dtype=torch.float32, device=dev)
        self.dlog = torch.zeros((batch, self.P), dtype=torch.float32, device=dev)
        self.doff = torch.zeros((batch, self.P, 2), dtype=torch.float32, device=dev)
        self.dvid = torch.zeros((batch,), dtype=torch.float32, device=dev)
        self.dreco = None
        self.loss_cfg = (float(model.train_center_sample_radius), model.train_center_sample == 'radius',
                         float(model.train_label_smoothing), float(model.train_loss_weight),
                         float(model.video_loss_weight), float(model.loss_normalizer_momentum))
        self.graphs = None
        self._outs = None
        # everything queued so far on the caller's stream (buffer fills, the RCCL broadcast of the initial weights) must be
        # visible to the probe, which copies the parameters into the flat buffer on the engine stream
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self._probe_and_flatten()
        torch.cuda.current_stream().wait_stream(self.stream)

    # --------------------------------------------------------------------------------------------
    def _cut_hook(self, t):
        """Backbone callback at the stem output: phase boundary of the backward pass (module docstring)."""
        self._cut_out = t
        self._cut_in = t.detach().requires_grad_(True)
        return self._cut_in

    def _forward(self):
        m = self.model
        logits, offsets, pmask, cls_scores, _ = m.forward_features(self.x, self.mask, cut_hook=self._cut_hook)
        reco = m._front_extra if self.has_reco else None
        return logits, offsets, pmask, cls_scores, reco

    def _phase_seeds(self, outs, grads):
        """[(tensors, grad seeds)] per backward phase, in execution order.  `grads` = (dlog, doff, dvid, dreco)."""
        logits, offsets, pmask, cls_scores, reco = outs
        dlog, doff, dvid, dreco = grads
        front = [] if not cls_scores.requires_grad else [(cls_scores, dvid.view_as(cls_scores))]
        if reco is not None:
            front.append((reco, dreco))
        phases = []
        if front and reco is None:
            phases.append(("front", front, None))          # independent of the backbone: first, its bucket travels longest
        phases.append(("post", [(logits, dlog), (offsets, doff)], None))
        # the stem output's gradient exists only after phase "post": fetched lazily
        phases.append(("pre", (front if reco is not None else []), "cut"))
        return phases

    def _run_phase(self, phase, retain):
        name, seeds, extra = phase
        tensors = [t for t, _ in seeds]
        grads = [g for _, g in seeds]
        if extra == "cut":
            tensors.append(self._cut_out)
            grads.append(self._cut_in.grad)
        torch.autograd.backward(tensors, grads, retain_graph=retain)

    # --------------------------------------------------------------------------------------------
    def _probe_and_flatten(self):
        """One eager probe step (on the engine stream) finds, per backward phase, the parameters autograd reaches; the
        rest are dead.  Live parameters are re-pointed into flat buffers laid out [bucket 0 | bucket 1 | ...], each bucket
        [decay | no_decay], so the optimiser is one launch and every bucket's all-reduce is one contiguous range.  The
        static input buffers and the RNG streams are left exactly as they were found."""
        m = self.model
        saved = (self.x.clone(), self.gt.clone(), self.ngt.clone())
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(self.dev)
        norm_state = m._loss_norm_state.clone()
        self.has_reco = False
        try:
            self.gt[:, 0, 0], self.gt[:, 0, 1] = 10.0, 30.0
            self.ngt.fill_(1)
            self.x.normal_()
            for p in m.parameters():
                p.grad = None
            logits, offsets, pmask, cls_scores, _ = m.forward_features(self.x, self.mask, cut_hook=self._cut_hook)
            extra = m._front_extra
            # exp5: the Expansion output (`reco`) feeds the backbone AND the L1 reconstruction loss (av_fd_meta_arch.py:
            # 346-348,683-701); exp13's frame scores only enter a reported, gradient-free term
            self.has_reco = extra is not None and extra.requires_grad and tuple(extra.shape) == tuple(self.x.shape)
            reco = extra if self.has_reco else None
            ones = (torch.ones_like(logits), torch.ones_like(offsets), torch.ones(self.B, device=self.dev),
                    None if reco is None else torch.ones_like(reco))
            named = dict(m.named_parameters())
            seen, buckets = set(), []
            for ph in self._phase_seeds((logits, offsets, pmask, cls_scores, reco), ones):
                self._run_phase(ph, retain=False)
                new = [n for n, p in named.items() if p.grad is not None and n not in seen]
                seen.update(new)
                buckets.append((ph[0], new))
            torch.cuda.synchronize(self.dev)
        finally:
            self.x.copy_(saved[0]); self.gt.copy_(saved[1]); self.ngt.copy_(saved[2])
            m._loss_norm_state.copy_(norm_state)
            torch.set_rng_state(rng_cpu)
            torch.cuda.set_rng_state(rng_dev, self.dev)
            self._cut_out = self._cut_in = None
        self.dead = sorted(n for n in named if n not in seen)
        decay, _ = split_decay(m)
        decay = set(decay)
        order, seg_end, seg_wd, self.bucket_ranges = [], [], [], []
        off = 0
        for name, names in buckets:
            lo = off
            for grp, wd in (([n for n in names if n in decay], self.wd), ([n for n in names if n not in decay], 0.0)):
                for n in grp:
                    order.append((n, off))
                    off += _pad4(named[n].numel())            # 16-byte aligned views (float4 kernels)
                if seg_end and seg_wd[-1] == wd:
                    seg_end[-1] = off
                elif off > (seg_end[-1] if seg_end else 0):
                    seg_end.append(off)
                    seg_wd.append(wd)
            self.bucket_ranges.append((name, lo, off))          # one (possibly empty) range per backward phase
        total = off
        self.n_total = sum(named[n].numel() for n, _ in order)
        self.n_decay = sum(named[n].numel() for n, _ in order if n in decay)
        assert len(seg_end) <= 8 and seg_end[-1] == total
        self._seg = ((ctypes.c_int64 * len(seg_end))(*seg_end), (ctypes.c_float * len(seg_wd))(*seg_wd), len(seg_end))
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.dev)
        for n, o in order:
            p = named[n]
            k = p.numel()
            self.flat_p[o:o + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + k].view(p.shape)
            p.grad = self.flat_g[o:o + k].view(p.shape)
            if not self.deterministic:
                p.main_grad = p.grad        # backward kernels accumulate here directly (ops._main_grad)
        for n in self.dead:
            named[n].grad = None
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.ema = self.flat_p.clone()
        self.live_names = [n for n, _ in order]
        self._offsets = dict(order)
        if self.has_reco:
            self.dreco = torch.zeros_like(self.x)

    # --------------------------------------------------------------------------------------------
    def _stage_a(self):
        ops.interp_concat_batched(self.stream_bufs, self.x)
        # the mode a layer's forward is issued under also selects the arithmetic of its backward kernels (bit 1)
        # forward-time fusion decisions (ops.mlp_residual_fusable: the MLP node's backward is the K = 1 pair launch on the
        # main stream) must see the setting their backward will run under
        ops.WGRAD_SIDE_STREAM = self.wgrad_side
        try:
            with ops.gemm_mode(ops.current_gemm_mode() | (1 if self.fwd_bf16x3 else 0) | (2 if self.bwd_bf16x3 else 0)):
                outs = self._forward()
        finally:
            ops.WGRAD_SIDE_STREAM = False
        st = torch.cuda.current_stream().cuda_stream
        radius, center = self.loss_cfg[0], self.loss_cfg[1]
        check(lib.avdf_loc_assign(self.points.data_ptr(), outs[2].data_ptr(), self.gt.data_ptr(), self.ngt.data_ptr(),
                                  self.ws_loss.data_ptr(), self.counts[0:1].data_ptr(), self.B, self.P,
                                  int(self.gt.shape[1]), radius, int(center), st), "loc_assign")
        if self.has_reco:
            check(lib.avdf_count_real(self.ngt.data_ptr(), self.counts[1:2].data_ptr(), self.B, st), "count_real")
        self._outs = outs
        self._phases = self._phase_seeds(outs, (self.dlog, self.doff, self.dvid, self.dreco))

    def _stage_loss(self):
        logits, offsets, pmask, cls_scores, reco = self._outs
        m = self.model
        st = torch.cuda.current_stream().cuda_stream
        _, _, ls, lw, vw, mom = self.loss_cfg
        vid = cls_scores.reshape(-1)
        check(lib.avdf_loc_losses(logits.data_ptr(), offsets.data_ptr(), pmask.data_ptr(), self.ngt.data_ptr(),
                                  vid.data_ptr(), self.ws_loss.data_ptr(), self.counts[0:1].data_ptr(),
                                  m._loss_norm_state.data_ptr(), self.loss_out.data_ptr(), self.dlog.data_ptr(),
                                  self.doff.data_ptr(), self.dvid.data_ptr(),
                                  self.ws_loss[self.B * self.P * 3:].data_ptr(), self.B, self.P, ls, lw, vw, mom, st),
              "loc_losses")
        if reco is not None:
            # + mean |reco - x| over the real clips of the global batch, straight into final_loss and d(final)/d(reco)
            check(lib.avdf_l1_mean_dev(reco.data_ptr(), self.x.data_ptr(), self.ngt.data_ptr(),
                                       self.counts[1:2].data_ptr(), self.loss_out[5:6].data_ptr(),
                                       self.loss_out[3:4].data_ptr(), self.dreco.data_ptr(), self.ws_l1.data_ptr(), self.B,
                                       self.T * self.C, st), "l1_mean_dev")

    def _stage_bwd(self, i):
        ops.WGRAD_SIDE_STREAM = self.wgrad_side
        ops.begin_deferred_reductions()          # the chains' parameter partials of this phase are folded in one go
        try:
            self._run_phase(self._phases[i], retain=self.use_graph)
        except BaseException:
            ops.abort_deferred_reductions()      # the phase's own error surfaces, not a fold over half-written partials
            raise
        else:
            ops.flush_deferred_reductions()
        finally:
            ops.WGRAD_SIDE_STREAM = False
            ops.wgrad_join()

    def _stage_c(self):
        st = torch.cuda.current_stream().cuda_stream
        n = self.flat_p.numel()
        check(lib.avdf_sumsq_ex(self.flat_g.data_ptr(), self.gnorm_sq.data_ptr(), self.ws_sumsq.data_ptr(), n, 0, st), "sumsq")
        b1, b2 = self.betas
        ends, wds, nseg = self._seg
        check(lib.avdf_adamw_ema_seg(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                     self.exp_avg_sq.data_ptr(), self.ema.data_ptr(), self.gnorm_sq.data_ptr(),
                                     self.hp.data_ptr(), n, nseg, ends, wds, b1, b2, self.eps, float(self.clip),
                                     self.ema_decay, 1, st), "adamw")

    @property
    def n_phases(self):
        return len(self._phases) if getattr(self, "_phases", None) else len(self.bucket_ranges)

    # --------------------------------------------------------------------------------------------
    def capture(self):
        """Warm up eagerly (allocator, lazy init), then capture the graphs -- all on the engine stream."""
        if not self.use_graph:
            return
        # the warm-up passes execute real forward / backward work: training state they touch (the EMA of the positive
        # count that normalises the losses, av_fd_no_recon.py:645-648; the gradient buffer) is restored afterwards, so
        # capture() is invisible
        norm_state = self.model._loss_norm_state.clone()
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(self.dev)
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._stage_a()
                self._stage_loss()
                for i in range(len(self._phases)):
                    self._stage_bwd(i)
                self._outs = None
            self.flat_g.zero_()
        torch.cuda.synchronize(self.dev)
        if self.world > 1:
            torch.distributed.barrier(group=self.pg)       # no collective of the set-up phase is in flight during capture
        self.model._loss_norm_state.copy_(norm_state)
        torch.set_rng_state(rng_cpu)
        torch.cuda.set_rng_state(rng_dev, self.dev)
        pool = torch.cuda.graph_pool_handle()

        def grab(fns):
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed may query events while this thread captures
            with torch.cuda.graph(g, pool=pool, stream=s, capture_error_mode="thread_local"):
                for fn in fns:
                    fn()
            return g

        nph = len(self._phases)
        bwd = [(lambda i=i: self._stage_bwd(i)) for i in range(nph)]
        if self.world == 1:
            # no exchange between the stages: the whole step is ONE graph launch
            self.graphs = ([grab([self._stage_a, self._stage_loss] + bwd + [self._stage_c])], None, None)
        else:
            ga = grab([self._stage_a])
            gbs = [grab(([self._stage_loss] if i == 0 else []) + [bwd[i]]) for i in range(nph)]
            self.graphs = (ga, gbs, grab([self._stage_c]))
        with torch.cuda.stream(s):
            self.flat_g.zero_()                  # capture does not execute, but keep the invariant explicit
        torch.cuda.current_stream().wait_stream(s)

    def set_lr(self, lr):
        self.lr = lr

    def set_lr_schedule(self, schedule):
        """`schedule(it)` -> lr of 0-based iteration `it` (libs.utils.lr_schedulers.make_scheduler); evaluated on the
        host before every step and written to the device scalars the captured optimiser kernel reads."""
        self.lr_schedule = schedule

    def _push_hparams(self):
        """lr and Adam's bias corrections of THIS step -> device.  The copy is asynchronous and steps are queued back to
        back, so each step writes its own pinned slot; a slot is rewritten only after the copy that read it completed."""
        t = self.step_count
        slot = t % self.HP_SLOTS
        if self._hp_done[slot] is not None:
            self._hp_done[slot].synchronize()
        h = self._hp_host[slot]
        h[0] = self.lr
        h[1] = 1.0 - self.betas[0] ** t
        h[2] = 1.0 - self.betas[1] ** t
        self.hp.copy_(h, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hp_done[slot] = ev

    def _reduce_bucket(self, i):
        _, lo, hi = self.bucket_ranges[i]
        if hi == lo:
            return None
        return torch.distributed.all_reduce(self.flat_g[lo:hi], group=self.pg, async_op=self.overlap)

    def step(self):
        """One optimisation step on the data currently in the static input buffers."""
        sched = getattr(self, "lr_schedule", None)
        if sched is not None:
            self.lr = float(sched(self.step_count))
        self.step_count += 1
        cur = torch.cuda.current_stream()
        own = self.graphs is None                          # eager: the engine stream (see __init__); graphs: the caller's
        run_stream = self.stream if own else cur
        if own:
            self.stream.wait_stream(cur)                   # inputs written on the caller's stream are visible
        dist = self.world > 1
        with torch.cuda.stream(run_stream):
            self._push_hparams()
            graphs = self.graphs
            if graphs is not None and graphs[1] is None:
                for g in graphs[0]:
                    g.replay()                             # world size 1: no exchange between the graphs
            else:
                if graphs is not None:
                    graphs[0].replay()
                else:
                    self._stage_a()
                if dist:
                    dp.allreduce_sum_(self.counts, group=self.pg)
                works = []
                nph = len(self._phases)
                assert nph == len(self.bucket_ranges)
                for i in range(nph):
                    if graphs is not None:
                        graphs[1][i].replay()
                    else:
                        if i == 0:
                            self._stage_loss()
                        self._stage_bwd(i)
                    if dist:
                        works.append(self._reduce_bucket(i))
                ev0 = ev1 = None
                if dist and self.comm_events is not None:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                for w in works:
                    if w is not None and self.overlap:
                        w.wait()
                if ev0 is not None:
                    ev1.record()
                    self.comm_events = (ev0, ev1)
                if graphs is not None:
                    graphs[2].replay()
                else:
                    self._outs = None
                    self._stage_c()
        if own:
            cur.wait_stream(self.stream)

    def losses(self, reduce=True):
        """The step's loss terms.  Under data parallelism every term is this rank's share of a global sum (the
        normaliser is already global): reduce=True all-reduces them -- a collective, call it on every rank."""
        o = self.loss_out.detach().clone()
        if self.world > 1 and reduce:
            num_pos = o[4].clone()
            dp.allreduce_sum_(o, group=self.pg)
            o[4] = num_pos                                  # already the global count
        o = o.cpu()
        out = {'cls_loss': float(o[0]), 'reg_loss': float(o[1]), 'reco_cls_loss': float(o[2]),
               'final_loss': float(o[3]), 'num_pos': float(o[4])}
        if self.has_reco:
            out['reco_loss'] = float(o[5])
        return out

    def ema_state_dict(self):
        """EMA weights under the model's state-dict keys (what inference.py loads as 'state_dict_ema')."""
        named = dict(self.model.named_parameters())
        out = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        for n in self.live_names:
            k, off = named[n].numel(), self._offsets[n]
            out[n] = self.ema[off:off + k].view(named[n].shape).clone()
        return out


    # ---- checkpoint-compatible state (libs/utils/train_utils.py:43-56, train.py:115-133) -----------------
    def _optimizer_order(self):
        """Parameter order of the reference's optimiser: make_optimizer builds two groups, sorted(decay) then
        sorted(no_decay) (train_utils.py:122-127); torch.optim numbers parameters in that order."""
        decay, no_decay = split_decay(self.model)
        return sorted(decay), sorted(no_decay)

    def state_dict(self):
        """The reference's checkpoint dict (train_utils.py:43-56 saves {'epoch', 'state_dict', 'state_dict_ema',
        'scheduler', 'optimizer'}): `state_dict` / `state_dict_ema` under the model's keys, `optimizer` in
        torch.optim.AdamW.state_dict() layout -- integer parameter ids in make_optimizer's order, per-parameter
        {'step', 'exp_avg', 'exp_avg_sq'} for every parameter that has received a gradient (dead parameters have no
        entry, as in torch.optim), two param_groups.  Extra keys carry what the reference keeps in Python attributes:
        `loss_normalizer` (av_fd_no_recon.py:645-648), `step_count`, and the RNG states of the drop-path draws."""
        torch.cuda.synchronize(self.dev)
        named = dict(self.model.named_parameters())
        decay, no_decay = self._optimizer_order()
        ids = {n: i for i, n in enumerate(decay + no_decay)}
        step = torch.tensor(float(self.step_count))
        state = {}
        if self.step_count > 0:
            for n in self.live_names:
                k, off, shp = named[n].numel(), self._offsets[n], named[n].shape
                state[ids[n]] = {'step': step.clone(), 'exp_avg': self.exp_avg[off:off + k].view(shp).clone(),
                                 'exp_avg_sq': self.exp_avg_sq[off:off + k].view(shp).clone()}
        common = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, amsgrad=False, maximize=False, foreach=None,
                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True)
        groups = [dict(common, weight_decay=self.wd, params=[ids[n] for n in decay]),
                  dict(common, weight_decay=0.0, params=[ids[n] for n in no_decay])]
        return {'state_dict': {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                'state_dict_ema': self.ema_state_dict(),
                'optimizer': {'state': state, 'param_groups': groups},
                'step_count': self.step_count,
                'loss_normalizer': self.model._loss_norm_state.detach().clone(),
                'rng_state': {'cpu': torch.get_rng_state(), 'cuda': torch.cuda.get_rng_state(self.dev)}}

    def load_state_dict(self, ckpt, strict=True):
        """Inverse of state_dict(); also accepts a checkpoint written by the reference's train.py (optimizer state of a
        torch.optim.AdamW built by make_optimizer; DataParallel's `module.` prefix on the model keys is dropped).  Weights,
        EMA and Adam moments go straight into the flat device buffers the captured step reads, so a captured engine
        resumes without re-capture."""
        torch.cuda.synchronize(self.dev)
        named = dict(self.model.named_parameters())

        def unprefixed(sd):
            return {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}

        sd = unprefixed(ckpt['state_dict'])
        missing = [n for n in named if n not in sd]
        if strict and missing:
            raise KeyError(f"state_dict lacks {len(missing)} parameters, e.g. {missing[:3]}")
        with torch.no_grad():
            for n, p in named.items():              # live parameters are views of flat_p: copy_ writes through
                if n in sd:
                    p.copy_(sd[n].to(self.dev))
            bufs = dict(self.model.named_buffers())
            for k, v in sd.items():
                if k in bufs and k not in named:
                    bufs[k].copy_(v.to(self.dev))
            ema = unprefixed(ckpt['state_dict_ema']) if ckpt.get('state_dict_ema') is not None else sd
            for n in self.live_names:
                k, off = named[n].numel(), self._offsets[n]
                self.ema[off:off + k].copy_(ema[n].to(self.dev).reshape(-1))
            decay, no_decay = self._optimizer_order()
            order = decay + no_decay
            opt = ckpt.get('optimizer')
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            steps = set()
            if opt is not None:
                groups = opt['param_groups']
                assert [len(g['params']) for g in groups] == [len(decay), len(no_decay)], \
                    "optimizer param_groups do not match make_optimizer's decay / no-decay partition of this model"
                flat_ids = [i for g in groups for i in g['params']]
                for pos, n in enumerate(order):
                    st = opt['state'].get(flat_ids[pos])
                    if st is None or n not in self._offsets:
                        continue
                    k, off = named[n].numel(), self._offsets[n]
                    self.exp_avg[off:off + k].copy_(st['exp_avg'].to(self.dev).reshape(-1))
                    self.exp_avg_sq[off:off + k].copy_(st['exp_avg_sq'].to(self.dev).reshape(-1))
                    steps.add(int(float(st['step'])))
                assert len(steps) <= 1, f"per-parameter step counts differ: {sorted(steps)}"
                self.lr = float(groups[0]['lr'])
            self.step_count = int(ckpt.get('step_count', steps.pop() if steps else 0))
            if ckpt.get('loss_normalizer') is not None:
                self.model._loss_norm_state.copy_(torch.as_tensor(ckpt['loss_normalizer'], dtype=torch.float32).reshape(
                    self.model._loss_norm_state.shape).to(self.dev))
            self.flat_g.zero_()
        rng = ckpt.get('rng_state')
        if rng is not None:
            torch.set_rng_state(rng['cpu'])
            torch.cuda.set_rng_state(rng['cuda'], self.dev)
        torch.cuda.synchronize(self.dev)


class InferStep:
    """Batched eval forward + on-device decode/NMS captured in one hipGraph (B clips per launch).
    bf16x3=True opts into split-bf16 forward GEMMs (~5e-6 relative error, ~1.4x faster); the default is exact fp32."""

    def __init__(self, model, batch, streams_shapes=None, use_graph=True, bf16x3=False):
        self.model = model.eval()
        self.dev = next(model.parameters()).device
        self.B, self.T, self.C = batch, model.max_seq_len, model.input_dim
        dev = self.dev
        self.bf16x3 = bf16x3
        # fixed-shape mode: one [B, T_s, C_s] buffer per stream kind (K1 runs as one launch per kind inside the graph);
        # streams_shapes=None: clips of any length are resampled into `x` by `load_items` before the captured part
        self.stream_bufs = None
        if streams_shapes is not None:
            self.stream_bufs = [torch.zeros((batch,) + tuple(shp), dtype=torch.float32, device=dev) for shp in streams_shapes]
            self.streams = [[buf[b] for buf in self.stream_bufs] for b in range(batch)]
        self.x = torch.zeros((batch, self.T, self.C), dtype=torch.float32, device=dev)
        self.mask = torch.ones((batch, self.T), dtype=torch.float32, device=dev)
        lens = [self.T // s for s in model.fpn_strides]
        self.points = torch.cat(model.point_generator(lens), dim=0).contiguous()
        self.use_graph = use_graph
        self.graph = None
        self.out = None

    @torch.no_grad()
    def _run(self):
        m = self.model
        if self.stream_bufs is not None:
            ops.interp_concat_batched(self.stream_bufs, self.x)
        with ops.infer_mode(self.bf16x3 or None):
            logits, offsets, pmask, cls_scores, _ = m.forward_features(self.x, self.mask)
        segs, scores, pidx, count = ops.decode_nms(
            self.points, logits, offsets, pmask, float(m.test_pre_nms_thresh), float(m.test_duration_thresh),
            float(m.test_min_score), float(m.test_iou_threshold), float(m.test_voting_thresh), int(m.test_max_seg_num))
        self.out = (segs, scores, pidx, count, cls_scores)

    def capture(self):
        if not self.use_graph:
            return
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._run()
            self._run()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._run()

    @torch.no_grad()
    def load_items(self, items):
        """Dataset items (`streams` = raw per-modality [T_s, C_s] arrays, any lengths) -> rows of the static input: K1
        per clip (force_upsampling: every clip becomes max_seq_len rows, deepfake_video_audio.py:510-547).  Unused rows
        of a partial batch are zeroed."""
        assert self.stream_bufs is None and len(items) <= self.B
        for b, it in enumerate(items):
            ss = [torch.as_tensor(s, dtype=torch.float32).to(self.dev, non_blocking=True) for s in it['streams']]
            ops.interp_concat(ss, self.T, out=self.x[b])
        if len(items) < self.B:
            self.x[len(items):].zero_()

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._run()
        return self.out


def run_sharded_inference(model, dataset, batch=8, process_group=None, output_folder=None, use_graph=True, bf16x3=False):
    """BASELINE config 5's localisation leg: the clip list of an inference dataset sharded over the data-parallel ranks
    (the reference splits its test list into `deepfake_test_sub{i}.txt` files and starts one `inference.py sub_index`
    process per file: inference.py:121, libs/datasets/deepfake_video_audio.py:420-431), B clips per captured launch on
    every rank, no collective on the data path, and one final gather of the <= max_seg_num segments per clip.

    Returns on rank 0 (None elsewhere) the list `inference_one_epoch` dumps as `data_left.json`
    (libs/utils/train_utils.py:548-595): [{'video_id', 'video_cls', 'scores', 'segments'}], in dataset order; written to
    `output_folder/data_left.json` when a folder is given."""
    dist = torch.distributed
    world = 1 if process_group is None else dist.get_world_size(process_group)
    rank = 0 if process_group is None else dist.get_rank(process_group)
    lo, hi = dp.shard_range(len(dataset), rank, world)
    eng = InferStep(model, batch, None, use_graph=use_graph, bf16x3=bf16x3)
    eng.capture()
    mine = []
    for i0 in range(lo, hi, batch):
        items = [dataset[i] for i in range(i0, min(i0 + batch, hi))]
        eng.load_items(items)
        segs, scores, pidx, count, cls_scores = eng.step()
        segs, scores, count, vcls = segs.cpu(), scores.cpu(), count.cpu(), cls_scores.detach().cpu()
        for b, it in enumerate(items):
            n = int(count[b])
            s = segs[b, :n].clone()
            if n > 0:                                       # postprocessing, av_fd_no_recon.py:827-876
                s = (s * it['feat_stride'] + 0.5 * it['feat_num_frames']) / it['fps']
                s[s <= 0.0] *= 0.0
                s[s >= it['duration']] = s[s >= it['duration']] * 0.0 + it['duration']
            mine.append({'video_id': it['video_id'], 'video_cls': vcls[b].numpy().tolist(),
                         'scores': scores[b, :n].numpy().tolist(), 'segments': s.numpy().tolist()})
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=dist.get_global_rank(process_group, 0), group=process_group)
        if rank != 0:
            return None
        mine = [r for part in gathered for r in part]
    if output_folder is not None:
        os.makedirs(output_folder, exist_ok=True)
        with open(os.path.join(output_folder, 'data_left.json'), 'w', encoding='utf-8') as f:
            json.dump(mine, f, ensure_ascii=False, indent=4)
    return mine
