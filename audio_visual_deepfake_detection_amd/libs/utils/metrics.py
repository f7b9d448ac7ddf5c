"""Temporal-localisation evaluation (mAP / top-kx recall at tIoU thresholds), host side, numpy only.

Same API and arithmetic as the reference's `libs/utils/metrics.py` (`ANETdetection` :112-252,
`compute_average_precision_detection` :255-335, `compute_topkx_recall_detection` :338-401, `segment_iou` :411-435,
`interpolated_prec_rec` :438-447, JSON loaders :13-109) without pandas / joblib -- and therefore without the
`np.float` alias the reference still uses (:326-327; removed in numpy 1.24, so the reference's own evaluator raises on
this image's numpy 2.2).  Tables are plain dicts of equal-length numpy arrays with the reference's column names
('video-id', 't-start', 't-end', 'label', 'score'); tie orders follow the reference because the same
`argsort()[::-1]` calls are applied to the same arrays.  Pinned by tests/golden/metrics.npz (reference output with
`np.float` aliased to `float` in the generator only)."""
import json
import os

import numpy as np

COLS_GT = ('video-id', 't-start', 't-end', 'label')
COLS_PRED = COLS_GT + ('score',)


def _table(d, cols):
    out = {c: np.asarray(d[c]) for c in cols}
    n = len(out[cols[0]])
    assert all(len(v) == n for v in out.values()), "columns must have equal length"
    return out


def _take(t, idx):
    return {k: v[idx] for k, v in t.items()}


def _groups(keys):
    """value -> indices (first-appearance order, like a pandas groupby on an unsorted key)."""
    g = {}
    for i, k in enumerate(keys.tolist() if hasattr(keys, "tolist") else keys):
        g.setdefault(k, []).append(i)
    return {k: np.asarray(v, dtype=np.int64) for k, v in g.items()}


def remove_duplicate_annotations(ants, tol=1e-3):
    valid_events = []
    for event in ants:
        s, e, l = event['segment'][0], event['segment'][1], event['label_id']
        valid = (e - s) >= tol
        for p in valid_events:
            if abs(s - p['segment'][0]) <= tol and abs(e - p['segment'][1]) <= tol and l == p['label_id']:
                valid = False
                break
        if valid:
            valid_events.append(event)
    return valid_events


def _label_id(v, label_offset):
    if isinstance(v, (tuple, list)):
        label_id = 0
        for i, x in enumerate(v[::-1]):
            label_id += label_offset ** i + int(x)
        return label_id
    return int(v)


def load_gt_seg_from_json(json_file, split=None, label='label_id', label_offset=0):
    with open(json_file, "r", encoding="utf8") as f:
        json_db = json.load(f)['database']
    vids, starts, stops, labels = [], [], [], []
    for k, v in json_db.items():
        if (split is not None) and v['subset'].lower() != split:
            continue
        ants = remove_duplicate_annotations(v['annotations'])
        vids += [k] * len(ants)
        for event in ants:
            starts.append(float(event['segment'][0]))
            stops.append(float(event['segment'][1]))
            labels.append(_label_id(event[label], label_offset))
    return _table({'video-id': np.asarray(vids, dtype=object), 't-start': np.asarray(starts, dtype=np.float64),
                   't-end': np.asarray(stops, dtype=np.float64), 'label': np.asarray(labels, dtype=np.int64)}, COLS_GT)


def load_pred_seg_from_json(json_file, label='label_id', label_offset=0):
    with open(json_file, "r", encoding="utf8") as f:
        json_db = json.load(f)['database']
    vids, starts, stops, labels, scores = [], [], [], [], []
    for k, v in json_db.items():
        vids += [k] * len(v)
        for event in v:
            starts.append(float(event['segment'][0]))
            stops.append(float(event['segment'][1]))
            labels.append(_label_id(event[label], label_offset))
            scores.append(float(event['scores']))
    return _table({'video-id': np.asarray(vids, dtype=object), 't-start': np.asarray(starts, dtype=np.float64),
                   't-end': np.asarray(stops, dtype=np.float64), 'label': np.asarray(labels, dtype=np.int64),
                   'score': np.asarray(scores, dtype=np.float64)}, COLS_PRED)


def segment_iou(target_segment, candidate_segments):
    tt1 = np.maximum(target_segment[0], candidate_segments[:, 0])
    tt2 = np.minimum(target_segment[1], candidate_segments[:, 1])
    inter = (tt2 - tt1).clip(0)
    union = (candidate_segments[:, 1] - candidate_segments[:, 0]) + (target_segment[1] - target_segment[0]) - inter
    return inter.astype(float) / union


def k_segment_iou(target_segments, candidate_segments):
    return np.stack([segment_iou(t, candidate_segments) for t in target_segments])


def interpolated_prec_rec(prec, rec):
    mprec = np.hstack([[0], prec, [0]])
    mrec = np.hstack([[0], rec, [1]])
    for i in range(len(mprec) - 1)[::-1]:
        mprec[i] = max(mprec[i], mprec[i + 1])
    idx = np.where(mrec[1::] != mrec[0:-1])[0] + 1
    return np.sum((mrec[idx] - mrec[idx - 1]) * mprec[idx])


def compute_average_precision_detection(ground_truth, prediction, tiou_thresholds=np.linspace(0.1, 0.5, 5)):
    ap = np.zeros(len(tiou_thresholds))
    npred = len(prediction['score']) if prediction is not None else 0
    if npred == 0:
        return ap
    ngt = len(ground_truth['t-start'])
    npos = float(ngt)
    lock_gt = np.ones((len(tiou_thresholds), ngt)) * -1
    sort_idx = prediction['score'].argsort()[::-1]
    prediction = _take(prediction, sort_idx)
    tp = np.zeros((len(tiou_thresholds), npred))
    fp = np.zeros((len(tiou_thresholds), npred))
    gt_by_vid = _groups(ground_truth['video-id'])
    gt_seg = np.stack([ground_truth['t-start'], ground_truth['t-end']], 1).astype(np.float64)
    pred_seg = np.stack([prediction['t-start'], prediction['t-end']], 1).astype(np.float64)
    for idx in range(npred):
        gidx = gt_by_vid.get(prediction['video-id'][idx])
        if gidx is None:
            fp[:, idx] = 1
            continue
        tiou_arr = segment_iou(pred_seg[idx], gt_seg[gidx])
        tiou_sorted_idx = tiou_arr.argsort()[::-1]
        for tidx, tiou_thr in enumerate(tiou_thresholds):
            for jdx in tiou_sorted_idx:
                if tiou_arr[jdx] < tiou_thr:
                    fp[tidx, idx] = 1
                    break
                if lock_gt[tidx, gidx[jdx]] >= 0:
                    continue
                tp[tidx, idx] = 1
                lock_gt[tidx, gidx[jdx]] = idx
                break
            if fp[tidx, idx] == 0 and tp[tidx, idx] == 0:
                fp[tidx, idx] = 1
    tp_cumsum = np.cumsum(tp, axis=1).astype(float)
    fp_cumsum = np.cumsum(fp, axis=1).astype(float)
    recall_cumsum = tp_cumsum / npos
    precision_cumsum = tp_cumsum / (tp_cumsum + fp_cumsum)
    for tidx in range(len(tiou_thresholds)):
        ap[tidx] = interpolated_prec_rec(precision_cumsum[tidx, :], recall_cumsum[tidx, :])
    return ap


def compute_topkx_recall_detection(ground_truth, prediction, tiou_thresholds=np.linspace(0.1, 0.5, 5), top_k=(1, 5)):
    npred = len(prediction['score']) if prediction is not None else 0
    if npred == 0:
        return np.zeros((len(tiou_thresholds), len(top_k)))
    tp = np.zeros((len(tiou_thresholds), len(top_k)))
    n_gts = 0
    gt_by_vid = _groups(ground_truth['video-id'])
    pred_by_vid = _groups(prediction['video-id'])
    gt_seg = np.stack([ground_truth['t-start'], ground_truth['t-end']], 1).astype(np.float64)
    pred_seg = np.stack([prediction['t-start'], prediction['t-end']], 1).astype(np.float64)
    for vid, gidx in gt_by_vid.items():
        n_gts += len(gidx)
        pidx = pred_by_vid.get(vid)
        if pidx is None:
            continue
        score_sort_idx = prediction['score'][pidx].argsort()[::-1]
        top_kx_idx = score_sort_idx[:max(top_k) * len(gidx)]
        tiou_arr = k_segment_iou(pred_seg[pidx][top_kx_idx], gt_seg[gidx])
        for tidx, tiou_thr in enumerate(tiou_thresholds):
            for kidx, k in enumerate(top_k):
                tiou = tiou_arr[:k * len(gidx)]
                tp[tidx, kidx] += ((tiou >= tiou_thr).sum(axis=0) > 0).sum()
    return tp / n_gts


class ANETdetection(object):
    """mAP / recall evaluator (reference `ANETdetection`, metrics.py:112-252)."""

    def __init__(self, ant_file, split=None, tiou_thresholds=np.linspace(0.1, 0.5, 5), top_k=(1, 5), label='label_id',
                 label_offset=0, num_workers=8, dataset_name=None):
        self.tiou_thresholds = tiou_thresholds
        self.top_k = top_k
        self.ap = None
        self.num_workers = num_workers          # kept for signature compatibility; evaluation is single-process
        if isinstance(ant_file, dict):          # in-memory ground truth (tests, synthetic data)
            self.dataset_name = dataset_name or "in-memory"
            self.ground_truth = _table(ant_file, COLS_GT)
        else:
            self.dataset_name = dataset_name if dataset_name is not None else os.path.basename(ant_file).replace('.json', '')
            self.ground_truth = load_gt_seg_from_json(ant_file, split=split, label=label, label_offset=label_offset)
        self.split = split
        self.activity_index = {j: i for i, j in enumerate(sorted(np.unique(self.ground_truth['label']).tolist()))}
        self.ground_truth['label'] = self._remap(self.ground_truth['label'])

    def _remap(self, labels):
        return np.asarray([self.activity_index.get(int(l), int(l)) for l in labels], dtype=np.int64)

    def _per_class(self, fn, preds, **kw):
        gt_by_label = _groups(self.ground_truth['label'])
        pred_by_label = _groups(preds['label'])
        res = []
        for label_name, cidx in self.activity_index.items():
            gt = _take(self.ground_truth, gt_by_label[cidx])
            pidx = pred_by_label.get(cidx)
            if pidx is None:
                print('Warning: No predictions of label \'%s\' were provdied.' % label_name)
            res.append(fn(ground_truth=gt, prediction=None if pidx is None else _take(preds, pidx),
                          tiou_thresholds=self.tiou_thresholds, **kw))
        return res

    def evaluate(self, preds, verbose=True):
        if isinstance(preds, str) and os.path.isfile(preds):
            preds = load_pred_seg_from_json(preds)
        else:
            conv = lambda v: v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            preds = _table({c: (np.asarray(preds[c], dtype=object) if c == 'video-id' else conv(preds[c])) for c in COLS_PRED},
                           COLS_PRED)
        preds['label'] = self._remap(preds['label'])
        self.ap = np.zeros((len(self.tiou_thresholds), len(self.activity_index)))
        self.recall = np.zeros((len(self.tiou_thresholds), len(self.top_k), len(self.activity_index)))
        aps = self._per_class(compute_average_precision_detection, preds)
        recs = self._per_class(compute_topkx_recall_detection, preds, top_k=self.top_k)
        for i, cidx in enumerate(self.activity_index.values()):
            self.ap[:, cidx] = aps[i]
            self.recall[..., cidx] = recs[i]
        mAP = self.ap.mean(axis=1)
        mRecall = self.recall.mean(axis=2)
        average_mAP = mAP.mean()
        if verbose:
            print('[RESULTS] Action detection results on {:s}.'.format(self.dataset_name))
            block = ''
            for tiou, tiou_mAP, tiou_mRecall in zip(self.tiou_thresholds, mAP, mRecall):
                block += '\n|tIoU = {:.2f}: '.format(tiou)
                block += 'mAP = {:>4.2f} (%) '.format(tiou_mAP * 100)
                for idx, k in enumerate(self.top_k):
                    block += 'Recall@{:d}x = {:>4.2f} (%) '.format(k, tiou_mRecall[idx] * 100)
            print(block)
            print('Average mAP: {:>4.2f} (%)'.format(average_mAP * 100))
        return mAP, average_mAP, mRecall
