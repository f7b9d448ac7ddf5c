"""Importable stand-in for the reference's pybind extension `nms_1d_cpu` (libs/utils/setup.py builds it from
libs/utils/csrc/nms_cpu.cpp; libs/utils/nms.py:5 imports it): the same two functions with the same argument names, served
by the host entry points of libavdf_hip.so.  Code written against the reference's module --

    import nms_1d_cpu
    inds = nms_1d_cpu.nms(segs, scores, iou_threshold=0.5)
    inds = nms_1d_cpu.softnms(segs, scores, dets, iou_threshold=0.5, sigma=0.5, min_score=0.001, method=2)

-- runs unchanged after `from audio_visual_deepfake_detection_amd.libs.utils import nms_1d_cpu` (or with this directory
on sys.path)."""
import ctypes

import numpy as np
import torch

from ..._lib import lib, check


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)


def nms(segs, scores, iou_threshold):
    """nms_cpu.cpp:19-58: indices (int64 tensor) of the kept segments in descending-score order."""
    s, c = _np32(segs).reshape(-1, 2), _np32(scores)
    n = c.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    nk = ctypes.c_int64(0)
    check(lib.avdf_nms1d_host(s.ctypes.data if n else None, c.ctypes.data if n else None, n, float(iou_threshold),
                              keep.ctypes.data, ctypes.addressof(nk)), "nms1d_host")
    return torch.from_numpy(keep[:nk.value].copy())


def softnms(segs, scores, dets, iou_threshold, sigma, min_score, method):
    """nms_cpu.cpp:60-150: fills dets[:k] = (start, end, decayed score) in emission order IN PLACE (as the extension does)
    and returns the k original indices."""
    s, c = _np32(segs).reshape(-1, 2), _np32(scores)
    n = c.shape[0]
    out = np.zeros((max(n, 1), 3), np.float32)
    inds = np.zeros(max(n, 1), np.int64)
    nk = ctypes.c_int64(0)
    check(lib.avdf_softnms1d_host(s.ctypes.data if n else None, c.ctypes.data if n else None, n, float(iou_threshold),
                                  float(sigma), float(min_score), int(method), out.ctypes.data, inds.ctypes.data,
                                  ctypes.addressof(nk)), "softnms1d_host")
    k = nk.value
    if k:
        dets[:k].copy_(torch.from_numpy(out[:k].copy()))
    return torch.from_numpy(inds[:k].copy())
