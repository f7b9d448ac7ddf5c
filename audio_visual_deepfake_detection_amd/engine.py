"""Host runtime for one training step of the UMMAFormer path on one MI355X (one process per GPU).

The reference's step is `losses = model(video_list); losses['final_loss'].backward();
clip_grad_norm_; optimizer.step(); scheduler.step(); model_ema.update()` (libs/utils/train_utils.py:
302-316) driven from Python with one aten launch per op.  Here the same step is three hipGraphs over
hand-written kernels, with the two data-parallel exchanges between them:

    graph A : K1 interpolation+concat of the raw feature streams -> network forward -> label assignment
    (DP)    : all-reduce(sum) of the positive-point count            [1 float, RCCL]
    graph B : fused losses (+ their gradients) -> network backward into ONE flat gradient buffer
    (DP)    : all-reduce(sum) of the flat gradient buffer             [RCCL over xGMI]
    graph C : grad-norm -> clipped AdamW -> weight EMA, one pass over flat buffers

Semantics kept from the reference: the focal / DIoU sums are divided by the EMA of the GLOBAL number
of positives (av_fd_no_recon.py:645-648,662-667), so gradients are SUM-reduced across ranks, not
averaged; parameters that never receive a gradient (interpolator.expansion.*, backbone.hh_branch.<last>)
are skipped by the optimiser exactly like torch.optim skips `grad is None`; weight decay applies to
conv/linear weights only (train_utils.py:73-111).
"""
import math

import torch

from . import ops
from ._lib import lib, check


def split_decay(model):
    """(decay, no_decay) parameter names with make_optimizer's rules (train_utils.py:73-111)."""
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


class TrainStep:
    """Owns the flat parameter / gradient / optimiser-state buffers and the captured step."""

    def __init__(self, model, batch, streams_shapes, lr=1e-3, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8,
                 clip_grad_l2norm=1.0, ema_decay=0.999, use_graph=True, process_group=None, max_gt=8, wgrad_side=False, fwd_bf16x3=False):
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
        self.step_count = 0
        dev = self.dev

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

        self._find_dead_and_flatten()

        self.num_pos = torch.zeros(1, dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(5, dtype=torch.float32, device=dev)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.hp = torch.zeros(3, dtype=torch.float32, device=dev)
        self._hp_host = torch.zeros(3, dtype=torch.float32).pin_memory()
        self.ws_loss = torch.empty(int(lib.avdf_loc_losses_ws(batch, self.P)), dtype=torch.float32, device=dev)
        self.ws_sumsq = torch.empty(int(lib.avdf_sumsq_ws()), dtype=torch.float32, device=dev)
        self.dlog = torch.zeros((batch, self.P), dtype=torch.float32, device=dev)
        self.doff = torch.zeros((batch, self.P, 2), dtype=torch.float32, device=dev)
        self.dvid = torch.zeros((batch,), dtype=torch.float32, device=dev)
        self.loss_cfg = (float(model.train_center_sample_radius), model.train_center_sample == 'radius',
                         float(model.train_label_smoothing), float(model.train_loss_weight),
                         float(model.video_loss_weight), float(model.loss_normalizer_momentum))
        self.graphs = None
        self._outs = None

    # --------------------------------------------------------------------------------------------
    def _find_dead_and_flatten(self):
        """One eager probe step finds the parameters autograd never reaches; the rest are re-pointed
        into flat buffers [decay | no_decay] so the optimiser and the all-reduce are single launches."""
        m = self.model
        self.gt[:, 0, 0], self.gt[:, 0, 1] = 10.0, 30.0
        self.ngt.fill_(1)
        for p in m.parameters():
            p.grad = None
        logits, offsets, pmask, cls_scores, _ = m.forward_features(self.x.normal_(), self.mask)
        (logits.sum() + offsets.sum() + cls_scores.sum()).backward()
        torch.cuda.synchronize()
        named = dict(m.named_parameters())
        self.dead = sorted(n for n, p in named.items() if p.grad is None)
        decay, no_decay = split_decay(m)
        live_decay = [n for n in decay if n not in set(self.dead)]
        live_nodecay = [n for n in no_decay if n not in set(self.dead)]
        self.n_decay = sum(named[n].numel() for n in live_decay)
        self.n_total = self.n_decay + sum(named[n].numel() for n in live_nodecay)
        # keep every view 16-byte aligned (float4 kernels): pad each tensor to a multiple of 4 floats
        def padded(n):
            return (n + 3) // 4 * 4
        order = live_decay + live_nodecay
        sizes = [padded(named[n].numel()) for n in order]
        self.n_decay_padded = sum(sizes[:len(live_decay)])
        total = sum(sizes)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.dev)
        off = 0
        for n, sz in zip(order, sizes):
            p = named[n]
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            p.main_grad = p.grad            # backward kernels accumulate here directly (ops._main_grad)
            off += sz
        for n in self.dead:
            named[n].grad = None
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.ema = self.flat_p.clone()
        self.live_names = order

    # --------------------------------------------------------------------------------------------
    def _stage_a(self):
        m = self.model
        ops.interp_concat_batched(self.stream_bufs, self.x)
        with ops.gemm_mode(1 if self.fwd_bf16x3 else ops.GEMM_MODE):
            logits, offsets, pmask, cls_scores, _ = m.forward_features(self.x, self.mask)
        st = torch.cuda.current_stream().cuda_stream
        radius, center, ls, lw, vw, mom = self.loss_cfg
        check(lib.avdf_loc_assign(self.points.data_ptr(), pmask.data_ptr(), self.gt.data_ptr(), self.ngt.data_ptr(),
                                  self.ws_loss.data_ptr(), self.num_pos.data_ptr(), self.B, self.P,
                                  int(self.gt.shape[1]), radius, int(center), st), "loc_assign")
        self._outs = (logits, offsets, pmask, cls_scores)

    def _stage_b(self):
        logits, offsets, pmask, cls_scores = self._outs
        m = self.model
        st = torch.cuda.current_stream().cuda_stream
        radius, center, ls, lw, vw, mom = self.loss_cfg
        vid = cls_scores.reshape(-1)
        check(lib.avdf_loc_losses(logits.data_ptr(), offsets.data_ptr(), pmask.data_ptr(), self.ngt.data_ptr(),
                                  vid.data_ptr(), self.ws_loss.data_ptr(), self.num_pos.data_ptr(),
                                  m._loss_norm_state.data_ptr(), self.loss_out.data_ptr(), self.dlog.data_ptr(),
                                  self.doff.data_ptr(), self.dvid.data_ptr(),
                                  self.ws_loss[self.B * self.P * 3:].data_ptr(), self.B, self.P, ls, lw, vw, mom, st),
              "loc_losses")
        self.flat_g.zero_()
        ops.WGRAD_SIDE_STREAM = self.wgrad_side
        try:
            torch.autograd.backward([logits, offsets, cls_scores], [self.dlog, self.doff, self.dvid.view_as(cls_scores)],
                                    retain_graph=self.use_graph)
        finally:
            ops.WGRAD_SIDE_STREAM = False
            ops.wgrad_join()

    def _stage_c(self):
        st = torch.cuda.current_stream().cuda_stream
        self.gnorm_sq.zero_()
        n = self.flat_p.numel()
        check(lib.avdf_sumsq(self.flat_g.data_ptr(), self.gnorm_sq.data_ptr(), self.ws_sumsq.data_ptr(), n, st), "sumsq")
        b1, b2 = self.betas
        nd = self.n_decay_padded
        for lo, hi, wd in ((0, nd, self.wd), (nd, n, 0.0)):
            if hi > lo:
                check(lib.avdf_adamw_ema(self.flat_p[lo:hi].data_ptr(), self.flat_g[lo:hi].data_ptr(),
                                         self.exp_avg[lo:hi].data_ptr(), self.exp_avg_sq[lo:hi].data_ptr(),
                                         self.ema[lo:hi].data_ptr(), self.gnorm_sq.data_ptr(), self.hp.data_ptr(),
                                         hi - lo, b1, b2, self.eps, wd, float(self.clip), self.ema_decay, st), "adamw")

    # --------------------------------------------------------------------------------------------
    def capture(self):
        """Warm up eagerly on a side stream (allocator, lazy init), then capture the three graphs."""
        if not self.use_graph:
            return
        # the warm-up passes execute real forward / backward work: training state they touch (the EMA of the positive
        # count that normalises the losses, av_fd_no_recon.py:645-648) is restored afterwards, so capture() is invisible
        norm_state = self.model._loss_norm_state.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._stage_a()
                self._stage_b()
                self._outs = None
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.model._loss_norm_state.copy_(norm_state)
        pool = torch.cuda.graph_pool_handle()
        ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, pool=pool):
            self._stage_a()
        with torch.cuda.graph(gb, pool=pool):
            self._stage_b()
        with torch.cuda.graph(gc, pool=pool):
            self._stage_c()
        self.graphs = (ga, gb, gc)

    def set_lr(self, lr):
        self.lr = lr

    def set_lr_schedule(self, schedule):
        """`schedule(it)` -> lr of 0-based iteration `it` (libs.utils.lr_schedulers.make_scheduler); evaluated on the
        host before every step and written to the device scalar the captured optimiser kernel reads."""
        self.lr_schedule = schedule

    def step(self):
        """One optimisation step on the data currently in the static input buffers."""
        sched = getattr(self, "lr_schedule", None)
        if sched is not None:
            self.lr = float(sched(self.step_count))
        self.step_count += 1
        t = self.step_count
        self._hp_host[0] = self.lr
        self._hp_host[1] = 1.0 - self.betas[0] ** t
        self._hp_host[2] = 1.0 - self.betas[1] ** t
        self.hp.copy_(self._hp_host, non_blocking=True)
        if self.graphs is not None:
            ga, gb, gc = self.graphs
            ga.replay()
            if self.world > 1:
                torch.distributed.all_reduce(self.num_pos, group=self.pg)
            gb.replay()
            if self.world > 1:
                torch.distributed.all_reduce(self.flat_g, group=self.pg)
            gc.replay()
        else:
            self._stage_a()
            if self.world > 1:
                torch.distributed.all_reduce(self.num_pos, group=self.pg)
            self._stage_b()
            self._outs = None
            if self.world > 1:
                torch.distributed.all_reduce(self.flat_g, group=self.pg)
            self._stage_c()

    def losses(self):
        o = self.loss_out.detach().cpu()
        return {'cls_loss': float(o[0]), 'reg_loss': float(o[1]), 'reco_cls_loss': float(o[2]),
                'final_loss': float(o[3]), 'num_pos': float(o[4])}

    def ema_state_dict(self):
        """EMA weights under the model's state-dict keys (what inference.py loads as 'state_dict_ema')."""
        named = dict(self.model.named_parameters())
        out = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        off = 0
        for n in self.live_names:
            k = named[n].numel()
            out[n] = self.ema[off:off + k].view(named[n].shape).clone()
            off += (k + 3) // 4 * 4
        return out


class InferStep:
    """Batched eval forward + on-device decode/NMS captured in one hipGraph (B clips per launch)."""

    def __init__(self, model, batch, streams_shapes, use_graph=True):
        self.model = model.eval()
        self.dev = next(model.parameters()).device
        self.B, self.T, self.C = batch, model.max_seq_len, model.input_dim
        dev = self.dev
        # one [B, T_s, C_s] buffer per stream kind (K1 runs as one launch per kind); streams[b][s] are views
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
        ops.interp_concat_batched(self.stream_bufs, self.x)
        with ops.infer_mode():
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

    def step(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._run()
        return self.out
