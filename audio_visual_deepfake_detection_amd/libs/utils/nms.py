"""1-D NMS with the reference's Python API (libs/utils/nms.py:103-190), backed by the native host implementations in
libavdf_hip.so (`avdf_nms1d_host`, `avdf_softnms1d_host`, `avdf_batched_nms_host`, which replace the pybind module
`nms_1d_cpu` of libs/utils/csrc/nms_cpu.cpp).  The model itself does not call this: its decode + NMS + voting runs
batched on the device (`avdf_decode_nms`); this entry point exists for callers of the reference API that hold CPU
tensors.  All four (soft | hard) x (multiclass | class-agnostic) branches are available."""
import ctypes

import numpy as np
import torch

from ..._lib import lib, check


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def _hard_nms(segs, scores, cls_idxs, iou_threshold, min_score, max_num):
    """NMSop (nms.py:8-35): score filter, greedy NMS, cap; rows come back in descending-score order."""
    if min_score > 0:
        valid = scores > min_score
        segs, scores, cls_idxs = segs[valid], scores[valid], cls_idxs[valid]
    s, c = _np32(segs).reshape(-1, 2), _np32(scores)
    n = c.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    nk = ctypes.c_int64(0)
    check(lib.avdf_nms1d_host(s.ctypes.data if n else None, c.ctypes.data if n else None, n, float(iou_threshold),
                              keep.ctypes.data, ctypes.addressof(nk)), "nms1d_host")
    k = nk.value if max_num <= 0 else min(nk.value, int(max_num))
    idx = torch.from_numpy(keep[:k].copy())
    return segs[idx].contiguous(), scores[idx].contiguous(), cls_idxs[idx].contiguous()


def _soft_nms(segs, scores, cls_idxs, iou_threshold, sigma, min_score, method, max_num):
    """SoftNMSop (nms.py:38-64): rows in emission order with their decayed scores."""
    s, c = _np32(segs).reshape(-1, 2), _np32(scores)
    n = c.shape[0]
    dets = np.zeros((max(n, 1), 3), np.float32)
    inds = np.zeros(max(n, 1), np.int64)
    nk = ctypes.c_int64(0)
    check(lib.avdf_softnms1d_host(s.ctypes.data if n else None, c.ctypes.data if n else None, n, float(iou_threshold),
                                  float(sigma), float(min_score), int(method), dets.ctypes.data, inds.ctypes.data,
                                  ctypes.addressof(nk)), "softnms1d_host")
    k = nk.value if max_num <= 0 else min(nk.value, int(max_num))
    d = torch.from_numpy(dets[:k].copy())
    return d[:, :2].contiguous(), d[:, 2].contiguous(), cls_idxs[torch.from_numpy(inds[:k].copy())]


class NMSop(torch.autograd.Function):
    """The reference's class name and call form (libs/utils/nms.py:8-35): NMSop.apply(segs, scores, cls_idxs, iou_threshold,
    min_score, max_num) -> (segs, scores, cls_idxs) of the kept rows in descending-score order."""

    @staticmethod
    def forward(ctx, segs, scores, cls_idxs, iou_threshold, min_score, max_num):
        return _hard_nms(segs, scores, cls_idxs, iou_threshold, min_score, max_num)


class SoftNMSop(torch.autograd.Function):
    """libs/utils/nms.py:38-64: SoftNMSop.apply(segs, scores, cls_idxs, iou_threshold, sigma, min_score, method, max_num);
    method 0 = hard, 1 = linear, 2 = gaussian decay (csrc/nms_cpu.cpp:60-150)."""

    @staticmethod
    def forward(ctx, segs, scores, cls_idxs, iou_threshold, sigma, min_score, method, max_num):
        return _soft_nms(segs, scores, cls_idxs, iou_threshold, sigma, min_score, method, max_num)


def seg_voting(nms_segs, all_segs, all_scores, iou_threshold, score_offset=1.5):
    """Segment voting (nms.py:67-101): each kept segment becomes the (score * IoU)-weighted mean of all input
    segments with IoU >= iou_threshold.  (`score_offset` is accepted and, as in the reference, has no effect.)"""
    a, b = nms_segs[:, None, :], all_segs[None, :, :]
    inter = (torch.minimum(a[..., 1], b[..., 1]) - torch.maximum(a[..., 0], b[..., 0])).clamp(min=0)
    iou = inter / ((a[..., 1] - a[..., 0]) + (b[..., 1] - b[..., 0]) - inter)
    wgt = (iou >= iou_threshold).to(all_scores.dtype) * all_scores[None, :] * iou
    wgt = wgt / wgt.sum(dim=1, keepdim=True)
    return wgt @ all_segs


def batched_nms(segs, scores, cls_idxs, iou_threshold, min_score, max_seg_num, use_soft_nms=True, multiclass=True,
                sigma=0.5, voting_thresh=0.75):
    if segs.shape[0] == 0:
        return torch.zeros([0, 2]), torch.zeros([0, ]), torch.zeros([0, ], dtype=cls_idxs.dtype)
    segs, scores, cls_idxs = segs.detach().cpu().float(), scores.detach().cpu().float(), cls_idxs.detach().cpu()

    def one(s, c, l):
        if use_soft_nms:
            return _soft_nms(s, c, l, iou_threshold, sigma, min_score, 2, max_seg_num)
        return _hard_nms(s, c, l, iou_threshold, min_score, max_seg_num)

    if multiclass:
        parts = [one(segs[cls_idxs == cid], scores[cls_idxs == cid], cls_idxs[cls_idxs == cid]) for cid in torch.unique(cls_idxs)]
        new_segs, new_scores, new_cls = (torch.cat([p[i] for p in parts]) for i in range(3))
    elif not use_soft_nms:
        # the shipped configs' branch, fused natively: filter -> NMS -> cap -> voting
        s, c = _np32(segs).reshape(-1, 2), _np32(scores)
        n = c.shape[0]
        cap = int(max_seg_num) if max_seg_num > 0 else n
        out_s = np.zeros((max(min(cap, n), 1), 2), np.float32)
        out_c = np.zeros((max(min(cap, n), 1),), np.float32)
        out_i = np.zeros((max(min(cap, n), 1),), np.int64)
        nk = ctypes.c_int64(0)
        check(lib.avdf_batched_nms_host(s.ctypes.data, c.ctypes.data, n, float(iou_threshold), float(min_score), cap,
                                        float(voting_thresh), out_s.ctypes.data, out_c.ctypes.data, out_i.ctypes.data,
                                        ctypes.addressof(nk)), "batched_nms_host")
        k = nk.value
        new_segs, new_scores = torch.from_numpy(out_s[:k].copy()), torch.from_numpy(out_c[:k].copy())
        new_cls = cls_idxs[torch.from_numpy(out_i[:k].copy())]
    else:
        new_segs, new_scores, new_cls = one(segs, scores, cls_idxs)
        if voting_thresh > 0:
            new_segs = seg_voting(new_segs, segs, scores, voting_thresh)
    idxs = new_scores.sort(descending=True, stable=True)[1]
    k = min(int(max_seg_num), new_segs.shape[0])
    return new_segs[idxs[:k]], new_scores[idxs[:k]], new_cls[idxs[:k]]
