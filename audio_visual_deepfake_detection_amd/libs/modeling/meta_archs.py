"""Audio-visual temporal-localisation meta-architecture on MI355X.

Drop-in for the reference's `AVLocPointTransformerRecoveryNoNormNorecon`
(libs/modeling/av_fd_no_recon.py:162-876): same constructor kwargs, same state-dict keys, same
`forward(video_list)` contract (training: dict of losses incl. 'final_loss'; eval: list of result
dicts with CPU tensors in seconds).  Everything between the H2D copy and the final D2H copy runs in
hand-written gfx950 kernels on time-major tensors.
"""
import math

import torch
from torch import nn

from .models import register_meta_arch, make_backbone, make_neck, make_generator
from .blocks import (MaskedConv1D, Scale, LayerNorm, DeepInterpolator, SegmentandCls, reset_mask_cache,
                     prime_mask_pyramid)
from ... import ops


_GAP_CACHE = {}


def gap_levels(fpn_feats, fpn_masks):
    """The FPN levels of a batch as ONE sequence [B, sum(T_l + 1), C]: an all-zero row after every level plays
    the zero padding of that level's k3 convs, so the heads (weights shared across levels, av_fd_no_recon.py:75-89,
    144-159) run as one launch per layer instead of one per level.  Returns (x, mask with zeros at the gap rows,
    per-level lengths, 0/1 row vector that is 0 only at the gap rows: LayerNorm+ReLU turn a zero row into relu(beta),
    so the gaps are re-zeroed after every norm -- masked real rows are NOT, as in the reference).  One HIP launch
    (ops.gap_concat); the head outputs are brought back to point order by ops.head_out."""
    B, _, C = fpn_feats[0].shape
    dev = fpn_feats[0].device
    lens = tuple(int(f.shape[1]) for f in fpn_feats)
    keep = gap_keep(B, lens, dev)
    x, gmask = ops.gap_concat(list(fpn_feats), list(fpn_masks))
    return x, gmask, lens, keep


def gap_keep(B, lens, dev):
    """0/1 row vector of the gapped level sequence (0 only at the gap rows), cached per (batch, lengths, device)."""
    key = (B, tuple(lens), dev)
    keep = _GAP_CACHE.get(key)
    if keep is None:
        off, rows = 0, []
        for n in lens:
            rows.append(torch.arange(off, off + n))
            off += n + 1
        keep = torch.zeros((B, off), dtype=torch.float32)
        keep[:, torch.cat(rows)] = 1.0
        keep = keep.to(dev)
        _GAP_CACHE[key] = keep
    return keep


def split_levels(y, lens):
    out, off = tuple(), 0
    for n in lens:
        out += (y[:, off:off + n],)
        off += n
    return out


class PtTransformerClsHead(nn.Module):
    """Shared classification head (av_fd_no_recon.py:13-89): 2 x [conv k3 -> LN -> ReLU] -> conv k3."""

    def __init__(self, input_dim, feat_dim, num_classes, prior_prob=0.01, num_layers=3, kernel_size=3,
                 act_layer=nn.ReLU, with_ln=False, empty_cls=[]):
        super().__init__()
        assert with_ln and act_layer is nn.ReLU and kernel_size == 3
        self.head, self.norm = nn.ModuleList(), nn.ModuleList()
        for idx in range(num_layers - 1):
            self.head.append(MaskedConv1D(input_dim if idx == 0 else feat_dim, feat_dim, kernel_size, stride=1,
                                          padding=kernel_size // 2, bias=(not with_ln)))
            self.norm.append(LayerNorm(feat_dim))
        self.cls_head = MaskedConv1D(feat_dim, num_classes, kernel_size, stride=1, padding=kernel_size // 2)
        if prior_prob > 0:
            torch.nn.init.constant_(self.cls_head.conv.bias, -(math.log((1 - prior_prob) / prior_prob)))
        if len(empty_cls) > 0:
            for idx in empty_cls:
                torch.nn.init.constant_(self.cls_head.conv.bias[idx], -(math.log((1 - 1e-6) / 1e-6)))

    def forward_gapped(self, x, gmask, keep):
        """All levels at once on the gapped sequence of `gap_levels`: [B, Tg, C] -> [B, Tg, nc]."""
        cur = x
        for idx in range(len(self.head)):
            cur, _ = self.head[idx](cur, gmask)
            cur = self.norm[idx](cur, relu=True, rowscale=keep)               # LN + ReLU, gap rows back to zero
        return ops.smallconv(cur, self.cls_head.conv.weight, self.cls_head.conv.bias, gmask)

    def forward(self, fpn_feats, fpn_masks):
        assert len(fpn_feats) == len(fpn_masks)
        x, gmask, lens, keep = gap_levels(fpn_feats, fpn_masks)
        return split_levels(ops.head_out(self.forward_gapped(x, gmask, keep), lens), lens)


class PtTransformerRegHead(nn.Module):
    """Shared regression head (av_fd_no_recon.py:92-159); the per-level Scale and the final ReLU are
    fused into the narrow-conv kernel."""

    def __init__(self, input_dim, feat_dim, fpn_levels, num_layers=3, kernel_size=3, act_layer=nn.ReLU,
                 with_ln=False):
        super().__init__()
        assert with_ln and act_layer is nn.ReLU and kernel_size == 3
        self.fpn_levels = fpn_levels
        self.head, self.norm = nn.ModuleList(), nn.ModuleList()
        for idx in range(num_layers - 1):
            self.head.append(MaskedConv1D(input_dim if idx == 0 else feat_dim, feat_dim, kernel_size, stride=1,
                                          padding=kernel_size // 2, bias=(not with_ln)))
            self.norm.append(LayerNorm(feat_dim))
        self.scale = nn.ModuleList([Scale() for _ in range(fpn_levels)])
        self.offset_head = MaskedConv1D(feat_dim, 2, kernel_size, stride=1, padding=kernel_size // 2)

    def forward_gapped(self, x, gmask, keep, lens):
        cur = x
        for idx in range(len(self.head)):
            cur, _ = self.head[idx](cur, gmask)
            cur = self.norm[idx](cur, relu=True, rowscale=keep)
        y = ops.smallconv(cur, self.offset_head.conv.weight, self.offset_head.conv.bias, gmask)
        # per-level Scale then ReLU (av_fd_no_recon.py:155-156) fused into the compaction to point order [B, P, 2]
        return ops.head_out(y, lens, relu=True, scales=[self.scale[l].scale for l in range(len(lens))])

    def forward(self, fpn_feats, fpn_masks):
        assert len(fpn_feats) == len(fpn_masks) == self.fpn_levels
        x, gmask, lens, keep = gap_levels(fpn_feats, fpn_masks)
        return split_levels(self.forward_gapped(x, gmask, keep, lens), lens)


@register_meta_arch("AVLocPointTransformerRecoveryNoNormNorecon")
class AVPtTransformerRecovery(nn.Module):
    """exp10 / exp12 model (configs_train/deepfake_exp10.yaml, configs_test/deepfake_exp12_test.yaml)."""

    BACKBONES = ('convHRLRFullResSelfAttTransformerRevised',)

    def __init__(self, backbone_type, fpn_type, backbone_arch, scale_factor, video_input_dim, audio_input_dim,
                 max_seq_len, max_buffer_len_factor, n_head, n_mha_win_size, embd_kernel_size, embd_dim, embd_with_ln,
                 fpn_dim, fpn_with_ln, fpn_start_level, head_dim, regression_range, head_num_layers,
                 head_kernel_size, head_with_ln, use_abs_pe, use_rel_pe, num_classes, train_cfg, test_cfg,
                 mlp_ratio=None):
        super().__init__()
        input_dim = video_input_dim + audio_input_dim
        self.input_dim = input_dim
        self.fpn_strides = [scale_factor ** i for i in range(fpn_start_level, backbone_arch[-1] + 1)]
        self.reg_range = regression_range
        assert len(self.fpn_strides) == len(self.reg_range)
        self.scale_factor = scale_factor
        self.num_classes = num_classes
        assert num_classes == 1, "the deepfake datasets are single-class (deepfake_video_audio.py:1536)"
        self.max_seq_len = max_seq_len
        if isinstance(n_mha_win_size, int):
            self.mha_win_size = [n_mha_win_size] * (1 + backbone_arch[-1])
        else:
            assert len(n_mha_win_size) == (1 + backbone_arch[-1])
            self.mha_win_size = n_mha_win_size
        max_div_factor = 1
        for s, w in zip(self.fpn_strides, self.mha_win_size):
            stride = s * (w // 2) * 2 if w > 1 else s
            assert max_seq_len % stride == 0, "max_seq_len must be divisible by fpn stride and window size"
            max_div_factor = max(max_div_factor, stride)
        self.max_div_factor = max_div_factor
        self.mlp_ratio = mlp_ratio

        self.train_center_sample = train_cfg['center_sample']
        assert self.train_center_sample in ['radius', 'none']
        self.train_center_sample_radius = train_cfg['center_sample_radius']
        self.train_loss_weight = train_cfg['loss_weight']
        assert self.train_loss_weight > 0, "auto-balanced loss weight (<=0) is unused by the shipped configs"
        self.train_cls_prior_prob = train_cfg['cls_prior_prob']
        self.train_dropout = train_cfg['dropout']
        self.train_droppath = train_cfg['droppath']
        self.train_label_smoothing = train_cfg['label_smoothing']

        self.test_pre_nms_thresh = test_cfg['pre_nms_thresh']
        self.test_pre_nms_topk = test_cfg['pre_nms_topk']
        self.test_iou_threshold = test_cfg['iou_threshold']
        self.test_min_score = test_cfg['min_score']
        self.test_max_seg_num = test_cfg['max_seg_num']
        self.test_nms_method = test_cfg['nms_method']
        assert self.test_nms_method in ['soft', 'hard', 'none']
        self.test_duration_thresh = test_cfg['duration_thresh']
        self.test_multiclass_nms = test_cfg['multiclass_nms']
        self.test_nms_sigma = test_cfg['nms_sigma']
        self.test_voting_thresh = test_cfg['voting_thresh']

        assert backbone_type in self.BACKBONES, (backbone_type, self.BACKBONES)
        if backbone_type == 'conv':                        # meta_archs.py:271-281
            self.backbone = make_backbone('conv', **{'n_in': input_dim, 'n_embd': embd_dim, 'n_embd_ks': embd_kernel_size,
                                                     'arch': backbone_arch, 'scale_factor': scale_factor,
                                                     'with_ln': embd_with_ln})
        else:
            self.backbone = make_backbone(backbone_type, **{
            'n_in': input_dim, 'n_embd': embd_dim, 'n_head': n_head, 'n_embd_ks': embd_kernel_size,
            'max_len': max_seq_len, 'arch': backbone_arch, 'mha_win_size': self.mha_win_size,
            'scale_factor': scale_factor, 'with_ln': embd_with_ln, 'attn_pdrop': 0.0,
            'proj_pdrop': self.train_dropout, 'path_pdrop': self.train_droppath, 'use_abs_pe': use_abs_pe,
            'use_rel_pe': use_rel_pe})
        if isinstance(embd_dim, (list, tuple)):
            embd_dim = sum(embd_dim)
        assert fpn_type in ['fpn', 'identity']
        self.neck = make_neck(fpn_type, **{
            'in_channels': [embd_dim] * (backbone_arch[-1] + 1), 'out_channel': fpn_dim,
            'scale_factor': scale_factor, 'start_level': fpn_start_level, 'with_ln': fpn_with_ln})
        self.point_generator = make_generator('point', **{
            'max_seq_len': max_seq_len * max_buffer_len_factor, 'fpn_strides': self.fpn_strides,
            'regression_range': self.reg_range})
        self.cls_head = PtTransformerClsHead(fpn_dim, head_dim, self.num_classes, kernel_size=head_kernel_size,
                                             prior_prob=self.train_cls_prior_prob, with_ln=head_with_ln,
                                             num_layers=head_num_layers, empty_cls=train_cfg['head_empty_cls'])
        self.reg_head = PtTransformerRegHead(fpn_dim, head_dim, len(self.fpn_strides), kernel_size=head_kernel_size,
                                             num_layers=head_num_layers, with_ln=head_with_ln)
        self._build_front(input_dim, embd_dim)

        # EMA of #foreground (av_fd_no_recon.py:321-322).  Kept on the device so a training step has no
        # host sync; `loss_normalizer` (python float in the reference) is exposed as a property.
        self.loss_normalizer_momentum = 0.9
        self.register_buffer("_loss_norm_state", torch.tensor([float(train_cfg['init_loss_norm'])]),
                             persistent=False)
        self.video_loss_weight = 0.1            # av_fd_no_recon.py:702
        self.num_pos_hook = None                # data-parallel: all-reduce of the positive count
        self.max_gt_per_clip = 16
        self.last_debug = {}

    # ------------------------------------------------------------------------------------------
    def _build_front(self, input_dim, embd_dim):
        """Video-level branch in front of the backbone (av_fd_no_recon.py:317)."""
        self.interpolator = DeepInterpolator(input_dim, embd_dim, norm=False)

    def _front(self, x, mask):
        """-> (video logits [B,1], reco_x for the backbone or None, extras for the losses)"""
        return self.interpolator(x, mask), None, None

    # ------------------------------------------------------------------------------------------
    @property
    def loss_normalizer(self):
        return float(self._loss_norm_state.item())

    @loss_normalizer.setter
    def loss_normalizer(self, v):
        self._loss_norm_state.fill_(float(v))

    @property
    def device(self):
        return list(set(p.device for p in self.parameters()))[0]

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def preprocessing(self, video_list, padding_val=0.0):
        """av_fd_no_recon.py:431-479, producing TIME-MAJOR device tensors: x [B, T, C], mask fp32 [B, T].

        Each item carries either the reference's 'feats' ([C, T_i] fp32, CPU or device), the fast-path
        'feats_tm' ([T_i, C] fp32 already on the device), or -- what this package's datasets produce --
        'streams' (raw per-modality [T_s, C_s] arrays) + 'crop': the K1 kernel resamples every stream to
        max_seq_len rows and concatenates them on the device (deepfake_video_audio.py:1722-1756), then the
        random crop of truncate_feats (data_utils.py:124) is a row slice.
        """
        dev = self.device
        assert dev.type == "cuda", "the MI355X path needs the model on a HIP device (no CPU fallback)"
        if 'streams' in video_list[0]:
            staged = []
            for v in video_list:
                ss = [torch.as_tensor(s, dtype=torch.float32).to(dev, non_blocking=True) for s in v['streams']]
                full = ops.interp_concat(ss, self.max_seq_len)
                st, ed = v['crop'] if v.get('crop') is not None else (0, self.max_seq_len)
                staged.append(dict(v, feats_tm=full[st:ed]))
            video_list = staged
        tm = 'feats_tm' in video_list[0]
        lens = [int(v['feats_tm'].shape[0] if tm else v['feats'].shape[-1]) for v in video_list]
        max_len = max(lens)
        if self.training:
            assert max_len <= self.max_seq_len, "Input length must be smaller than max_seq_len during training"
            max_len = self.max_seq_len
        else:
            assert len(video_list) == 1 or getattr(self, "allow_batched_eval", False), \
                "Only support batch_size = 1 during inference"
            if max_len <= self.max_seq_len:
                max_len = self.max_seq_len
            else:
                # longer than max_seq_len: pad to the next length every pyramid level / attention window divides
                # (av_fd_no_recon.py:458-463); needs max_buffer_len_factor > 1 for the point generator, as in the reference
                stride = self.max_div_factor
                max_len = (max_len + (stride - 1)) // stride * stride
        B, C = len(video_list), self.input_dim
        if tm:
            x = torch.full((B, max_len, C), padding_val, dtype=torch.float32, device=dev)
            for b, v in enumerate(video_list):
                x[b, :lens[b]].copy_(v['feats_tm'])
        else:
            ct = video_list[0]['feats'].new_full((B, C, max_len), padding_val)
            for b, v in enumerate(video_list):
                ct[b, :, :lens[b]].copy_(v['feats'])
            x = ops.transpose_last2(ct.to(dev, non_blocking=True))         # [B,C,T] -> [B,T,C] on device
        mask = (torch.arange(max_len)[None, :] < torch.as_tensor(lens)[:, None]).to(torch.float32).to(dev)
        return x, mask

    # ------------------------------------------------------------------------------------------
    def forward_features(self, x, mask, cut_hook=None):
        """Network body on time-major tensors: returns (logits [B,P], offsets [B,P,2], pmask [B,P],
        video logits [B,1], per-level lengths).  `cut_hook`: see the backbone's forward."""
        reset_mask_cache()
        prime_mask_pyramid(mask, len(self.fpn_strides) - 1, self.scale_factor)
        self._front_input = x
        cls_scores, reco_x, self._front_extra = self._front(x, mask)
        feats, masks = self._run_backbone(x, reco_x, mask, cut_hook)
        if hasattr(self.neck, "forward_gapped"):
            out = self.neck.forward_gapped(feats, masks)
        else:
            out = self.neck(feats, masks)
        if len(out) == 3:                      # fused FPN tail: already the gapped level sequence
            gx, gmask, lens = out
            keep = gap_keep(gx.shape[0], lens, gx.device)
        else:
            gx, gmask, lens, keep = gap_levels(*out)
        lens = list(lens)
        logits = ops.head_out(self.cls_head.forward_gapped(gx, gmask, keep), lens).squeeze(-1)
        offsets = self.reg_head.forward_gapped(gx, gmask, keep, lens)
        pmask = ops.head_out(gmask.unsqueeze(-1), lens).squeeze(-1)             # point-order mask = gapped mask compacted
        return logits, offsets, pmask, cls_scores, lens

    def _run_backbone(self, x, reco_x, mask, cut_hook=None):
        return self.backbone(x, reco_x, mask, cut_hook=cut_hook)

    def forward(self, video_list):
        if not self.training:
            with ops.infer_mode():             # exact fp32 unless bf16x3 was opted into (ops.INFER_BF16X3)
                return self._forward(video_list)
        return self._forward(video_list)

    def _forward(self, video_list):
        x, mask = self.preprocessing(video_list)
        logits, offsets, pmask, cls_scores, lens = self.forward_features(x, mask)
        points = torch.cat(self.point_generator(lens), dim=0)
        if self.training:
            gt, ngt = self._pack_gt(video_list, x.device)
            losses = self.losses(points, logits, offsets, pmask, cls_scores, gt, ngt)
            return self._extra_losses(losses, video_list)
        return self.inference(video_list, points, pmask, logits, offsets, cls_scores, lens)

    # ------------------------------------------------------------------------------------------
    def _pack_gt(self, video_list, dev):
        """`segments is None` marks a real clip (av_fd_no_recon.py:388-395): ngt = 0."""
        G = max([self.max_gt_per_clip] + [0 if v['segments'] is None else int(v['segments'].shape[0])
                                         for v in video_list])
        gt = torch.zeros((len(video_list), G, 2), dtype=torch.float32)
        ngt = torch.zeros((len(video_list),), dtype=torch.int32)
        for b, v in enumerate(video_list):
            if v['segments'] is not None:
                n = int(v['segments'].shape[0])
                assert n > 0, "a fake clip must carry at least one segment"
                gt[b, :n] = v['segments'].to(torch.float32)
                ngt[b] = n
        return gt.to(dev), ngt.to(dev)

    def losses(self, points, logits, offsets, pmask, cls_scores, gt, ngt):
        """label_points + losses (av_fd_no_recon.py:481-707) in two fused kernels."""
        cfg = (float(self.train_center_sample_radius), self.train_center_sample == 'radius',
               float(self.train_label_smoothing), float(self.train_loss_weight), float(self.video_loss_weight),
               float(self.loss_normalizer_momentum))
        out = ops.loc_losses(logits, offsets, cls_scores.reshape(-1), points, pmask, gt, ngt,
                             self._loss_norm_state, cfg, self.num_pos_hook)
        det = out.detach()
        self.last_debug['num_pos'] = det[4]
        return {'cls_loss': det[0], 'reg_loss': det[1], 'reco_cls_loss': det[2], 'final_loss': out[3]}

    def _extra_losses(self, losses, video_list):
        return losses

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def inference(self, video_list, points, pmask, logits, offsets, cls_scores, lens):
        """decode + NMS + voting on the device (av_fd_no_recon.py:709-825, nms.py), then the
        seconds conversion of postprocessing (av_fd_no_recon.py:827-876) on the host."""
        assert self.test_nms_method == 'hard' and not self.test_multiclass_nms, \
            "only the class-agnostic hard-NMS branch used by the shipped configs is implemented"
        assert self.test_pre_nms_topk >= max(lens), \
            "per-level top-k below the level length is not implemented (pre_nms_topk is 2000-5000 in every config)"
        segs, scores, pidx, count = ops.decode_nms(
            points, logits, offsets, pmask, float(self.test_pre_nms_thresh), float(self.test_duration_thresh),
            float(self.test_min_score), float(self.test_iou_threshold), float(self.test_voting_thresh),
            int(self.test_max_seg_num))
        segs, scores, pidx, count = segs.cpu(), scores.cpu(), pidx.cpu(), count.cpu()
        video_cls = cls_scores.detach().cpu()
        results = []
        for b, v in enumerate(video_list):
            n = int(count[b])
            s, sc = segs[b, :n].clone(), scores[b, :n].clone()
            if n > 0:
                s = (s * v['feat_stride'] + 0.5 * v['feat_num_frames']) / v['fps']
                s[s <= 0.0] *= 0.0
                s[s >= v['duration']] = s[s >= v['duration']] * 0.0 + v['duration']
            results.append({'video_id': v['video_id'], 'segments': s, 'scores': sc,
                            'labels': torch.zeros(n, dtype=torch.int64), 'video_cls': video_cls[b],
                            'kept_points': pidx[b, :n].to(torch.int64)})
        return results


@register_meta_arch("AVLocPointTransformerRecoveryNoNormNoreconTHE")
class AVPtTransformerRecoveryTHE(AVPtTransformerRecovery):
    """exp13 model (configs_train/deepfake_exp13.yaml; reference: libs/modeling/av_fd_no_recon2.py):
    `SegmentandCls` replaces `DeepInterpolator`; an extra frame-level focal loss `reco_seg_loss` is
    REPORTED but not part of `final_loss` (av_fd_no_recon2.py:718-733)."""

    def _build_front(self, input_dim, embd_dim):
        self.segmentandCls = SegmentandCls(input_dim, norm=False)

    def _front(self, x, mask):
        seg_scores, cls_scores = self.segmentandCls(x, mask)
        return cls_scores, None, seg_scores

    def _extra_losses(self, losses, video_list):
        seg_scores = self._front_extra.detach()
        idx = [b for b, v in enumerate(video_list) if v['segments'] is not None]
        T = seg_scores.shape[1]
        if idx:
            gt = torch.zeros((len(idx), T), dtype=torch.float32)
            for j, b in enumerate(idx):
                fl = video_list[b]['gt_frame_labels'].to(torch.float32).reshape(-1)
                gt[j, :fl.numel()] = fl
            s = ops.focal_loss_sum(seg_scores[idx].contiguous(), gt.to(seg_scores.device))
            losses['reco_seg_loss'] = (s / (self._loss_norm_state * 10.0)).reshape(())
        else:
            losses['reco_seg_loss'] = torch.zeros((), device=seg_scores.device)
        return losses


@register_meta_arch("AVLocPointTransformerRecoveryNoNorm")
class AVPtTransformerRecoveryNoNorm(AVPtTransformerRecovery):
    """exp5 model (configs_train/deepfake_exp5.yaml; reference: libs/modeling/av_fd_meta_arch.py): the
    Expansion output `reco_result` is the key source of the first cross-attention block (:346-348) and an
    L1 reconstruction loss on the REAL clips is added to `final_loss` (:683-686,701).  As in the reference,
    a batch without any real clip makes that mean -- and hence the loss -- NaN."""

    def _front(self, x, mask):
        cls_scores, reco = self.interpolator(x, mask, need_reco=True)
        return cls_scores, reco, reco

    def _extra_losses(self, losses, video_list):
        reco, x = self._front_extra, self._front_input
        real = [1.0 if v['segments'] is None else 0.0 for v in video_list]
        wb = torch.tensor(real, dtype=torch.float32).to(x.device)
        reco_loss = ops.l1_mean(reco, x.detach(), wb, int(sum(real))).reshape(())
        losses['reco_loss'] = reco_loss.detach()
        losses['final_loss'] = losses['final_loss'] + reco_loss
        return losses


@register_meta_arch("LocPointTransformer")
class PtTransformer(AVPtTransformerRecovery):
    """Plain single-stage localiser (reference: libs/modeling/meta_archs.py:162-700; the `model_name` default of
    libs/core/config.py): convTransformer backbone -> FPN -> heads, no video-level branch.  Training expects every clip
    to carry segments (meta_archs.py:346-348); losses / results have the reference's keys (no `reco_cls_loss`, no
    `video_cls`)."""

    BACKBONES = ('convTransformer', 'conv')

    def __init__(self, backbone_type, fpn_type, backbone_arch, scale_factor, max_seq_len, max_buffer_len_factor, n_head,
                 n_mha_win_size, embd_kernel_size, embd_dim, embd_with_ln, fpn_dim, fpn_with_ln, fpn_start_level, head_dim,
                 regression_range, head_num_layers, head_kernel_size, head_with_ln, use_abs_pe, use_rel_pe, num_classes,
                 train_cfg, test_cfg, input_dim=None, audio_input_dim=0, video_input_dim=None):
        # the reference's signature takes `input_dim` (meta_archs.py:172-173) while its config loader fills
        # `video_input_dim` / `audio_input_dim` (libs/core/config.py:151-152): both spellings are accepted here
        if input_dim is None:
            input_dim = int(video_input_dim) + int(audio_input_dim or 0)
        super().__init__(backbone_type, fpn_type, backbone_arch, scale_factor, input_dim, 0, max_seq_len,
                         max_buffer_len_factor, n_head, n_mha_win_size, embd_kernel_size, embd_dim, embd_with_ln, fpn_dim,
                         fpn_with_ln, fpn_start_level, head_dim, regression_range, head_num_layers, head_kernel_size,
                         head_with_ln, use_abs_pe, use_rel_pe, num_classes, train_cfg, test_cfg)
        self.video_loss_weight = 0.0

    def _build_front(self, input_dim, embd_dim):
        pass

    def _front(self, x, mask):
        return torch.zeros((x.shape[0], 1), dtype=torch.float32, device=x.device), None, None

    def _run_backbone(self, x, reco_x, mask, cut_hook=None):
        return self.backbone(x, mask, cut_hook=cut_hook)

    def _pack_gt(self, video_list, dev):
        assert all(v.get('segments') is not None for v in video_list), "GT action labels does not exist"
        return super()._pack_gt(video_list, dev)

    def _extra_losses(self, losses, video_list):
        return {k: losses[k] for k in ('cls_loss', 'reg_loss', 'final_loss')}

    def inference(self, video_list, points, pmask, logits, offsets, cls_scores, lens):
        results = super().inference(video_list, points, pmask, logits, offsets, cls_scores, lens)
        for r in results:
            r.pop('video_cls')
        return results


@register_meta_arch("AVLocPointTransformer")
class AVPtTransformer(PtTransformer):
    """Reference: libs/modeling/av_meta_arch.py:162-700 -- LocPointTransformer on the channel-concatenated audio-visual
    input (`input_dim + audio_input_dim`, av_meta_arch.py:198); same modules, state-dict keys, losses and outputs."""

    def __init__(self, *args, input_dim=None, audio_input_dim=0, video_input_dim=None, **kw):
        base = input_dim if input_dim is not None else video_input_dim
        super().__init__(*args, input_dim=int(base) + int(audio_input_dim or 0), audio_input_dim=0, **kw)
