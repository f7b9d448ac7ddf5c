"""1-D NMS with the reference's Python API (libs/utils/nms.py:103-190), backed by the native host
implementation in libavdf_hip.so (`avdf_batched_nms_host`, which replaces the pybind module `nms_1d_cpu` of
libs/utils/csrc/nms_cpu.cpp).  The model itself does not call this: its decode+NMS+voting runs batched on the
device (`avdf_decode_nms`); this entry point exists for callers of the reference API that hold CPU tensors."""
import ctypes

import numpy as np
import torch

from ..._lib import lib, check


def batched_nms(segs, scores, cls_idxs, iou_threshold, min_score, max_seg_num, use_soft_nms=True, multiclass=True,
                sigma=0.5, voting_thresh=0.75):
    if segs.shape[0] == 0:
        return torch.zeros([0, 2]), torch.zeros([0, ]), torch.zeros([0, ], dtype=cls_idxs.dtype)
    if use_soft_nms or multiclass:
        raise NotImplementedError("only the class-agnostic hard-NMS branch used by the shipped configs is implemented")
    s = np.ascontiguousarray(segs.detach().cpu().numpy(), dtype=np.float32)
    c = np.ascontiguousarray(scores.detach().cpu().numpy(), dtype=np.float32)
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
    idx = torch.from_numpy(out_i[:k].copy())
    return torch.from_numpy(out_s[:k].copy()), torch.from_numpy(out_c[:k].copy()), cls_idxs.detach().cpu()[idx]
