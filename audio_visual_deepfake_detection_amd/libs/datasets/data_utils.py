"""Host-side helpers of the dataset plugins (reference: libs/datasets/data_utils.py)."""
import os
import random

import numpy as np
import torch


def trivial_batch_collator(batch):
    return batch


def worker_init_reset_seed(worker_id):
    seed = torch.initial_seed() % 2 ** 31
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def truncate_window(feat_len, segments, labels, max_seq_len, trunc_thresh, offset, crop_ratio=None,
                    max_num_trials=200):
    """Random temporal crop of `truncate_feats` (data_utils.py:45-134) WITHOUT touching the features: returns
    (st, ed, segments, labels) so the crop can be applied on the device after the K1 interpolation.
    Draws from python's `random` in the same order as the reference (one randint for the crop length, one
    per trial for the start), keeps a window only if >= 1 segment overlaps it by >= trunc_thresh."""
    if feat_len <= max_seq_len:
        if crop_ratio is None:
            return 0, feat_len, segments, labels
        max_seq_len = random.randint(max(round(crop_ratio[0] * feat_len), 1), min(round(crop_ratio[1] * feat_len), feat_len))
        if feat_len == max_seq_len:
            return 0, feat_len, segments, labels
    num_segs = segments.shape[0]
    for _ in range(max_num_trials):
        st = random.randint(0, feat_len - max_seq_len)
        ed = st + max_seq_len
        window = torch.as_tensor([st, ed], dtype=torch.float32)[None].repeat(num_segs, 1)
        left = torch.maximum(window[:, 0] - offset, segments[:, 0])
        right = torch.minimum(window[:, 1] + offset, segments[:, 1])
        inter = (right - left).clamp(min=0)
        area = torch.abs(segments[:, 1] - segments[:, 0])
        seg_idx = (inter / area) >= trunc_thresh
        if seg_idx.sum().item() > 0:
            break
    new_segments = torch.stack((left[seg_idx], right[seg_idx]), dim=1) - st
    return st, ed, new_segments, labels[seg_idx].clone()
