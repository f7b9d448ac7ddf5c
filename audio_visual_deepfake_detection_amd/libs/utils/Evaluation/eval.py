"""Detections -> ANet-style json -> AP, and the challenge submission files.

Reference: libs/utils/Evaluation/eval.py (`post_process_multi` :119-145 with `detection_thread` :99-117,
`evaluation_detection` :82-96, `run_evaluation` :148-165) and generate_results.ipynb (the README's "confidence below 0.2
-> empty list" rule).  Tables are dicts of equal-length arrays (or anything with the same attribute / key names, e.g. a
pandas DataFrame); no joblib pool: a video's rows are one numpy slice."""
import json

import numpy as np

from .eval_detection import ANETdetection


def _col(t, name):
    v = t[name] if isinstance(t, dict) else getattr(t, name)
    return np.asarray(getattr(v, "values", v))


def post_process_multi(pred_data, output_file, cls_score_file=None):
    """eval.py:119-145: per video the first <= 100 rows (table order), label 'Fake', score x the best video-level score of
    `cls_score_file` (1 without it), segment start clamped at 0 -> {"version", "results", "external_data"} json."""
    names = _col(pred_data, 'video_name')
    score, xmin, xmax = _col(pred_data, 'score'), _col(pred_data, 'xmin'), _col(pred_data, 'xmax')
    best_cls = None
    if cls_score_file is not None:
        with open(cls_score_file) as f:
            best_cls = json.load(f)
    rows = {}
    for i, v in enumerate(names.tolist()):
        rows.setdefault(v, []).append(i)
    detection_dict = {}
    for vid, idx in rows.items():
        best = float(np.max(best_cls[vid])) if best_cls is not None else 1.0
        detection_dict[vid] = [{"label": 'Fake', "score": float(score[j]) * best,
                                "segment": [max(0, float(xmin[j])), float(xmax[j])]} for j in idx[:100]]
    with open(output_file, "w") as out:
        json.dump({"version": "ANET v1.3, Lavdf", "results": detection_dict, "external_data": {}}, out)


def evaluation_detection(gt_filename, pred_filename, tious, subset):
    """eval.py:82-96: prints / appends the report line next to the prediction file, returns mean AP x 100."""
    det = ANETdetection(ground_truth_filename=gt_filename, prediction_filename=pred_filename, tiou_thresholds=tious,
                        subset=subset, verbose=True, check_status=False)
    det.evaluate()
    at = [f'mAP@{t:.2f} {m * 100:.3f}' for t, m in zip(det.tiou_thresholds, det.mAP)]
    results = f'Detection: average-mAP {det.average_mAP * 100:.3f} {" ".join(at)}'
    print(results)
    with open(pred_filename.replace('.json', '.txt'), 'a') as fobj:
        fobj.write(f'{results}\n')
    return np.mean(det.mAP) * 100


def run_evaluation(preds, ground_truth_file, proposal_file, dataset_name='', max_avg_nr_proposal=100,
                   tiou_thre=np.array([0.5, 0.75, 0.9, 0.95]), subset='test', cls_score_file=None):
    """eval.py:148-165, called by valid_one_epoch (train_utils.py:497) with the results dict it accumulated
    ('video-id', 't-start', 't-end', 'label', 'score')."""
    table = {'video_name': np.asarray(preds['video-id'], dtype=object), 'xmin': np.asarray(preds['t-start']),
             'xmax': np.asarray(preds['t-end']), 'label': np.asarray(preds['label']), 'score': np.asarray(preds['score'])}
    print("saving detection results...")
    post_process_multi(table, proposal_file, cls_score_file)
    print("evaluion detection results...")
    mAP = evaluation_detection(ground_truth_file, proposal_file, tiou_thre, subset)
    return mAP, 0.1


def format_submission(result_lists, min_score=0.2, cls_saturate=0.9):
    """generate_results.ipynb: `result_lists` = the `data_left*.json` lists of the inference shards
    (engine.run_sharded_inference / inference_one_epoch: video_id, video_cls, scores, segments), first occurrence of a
    video wins.  -> (classification lines 'video_id;prob' sorted by id, prob = sigmoid(video_cls) saturated to 1.0 above
    0.9; localisation dict video_id -> [[score, start, end], ...] keeping score > 0.2, [[0, 0, 0]] when none)."""
    seen, cls_lines, loc = set(), [], {}
    for data in result_lists:
        for item in data:
            vid = item['video_id']
            if vid in seen:
                continue
            seen.add(vid)
            logit = np.asarray(item['video_cls'], np.float32).reshape(-1)[0]
            prob = float(np.float32(1.0) / (np.float32(1.0) + np.exp(-logit, dtype=np.float32)))
            cls_lines.append([vid, str(1.0 if prob > cls_saturate else prob)])
            keep = [[s, item['segments'][i][0], item['segments'][i][1]] for i, s in enumerate(item['scores']) if s > min_score]
            loc[vid] = keep if keep else [[0, 0, 0]]
    cls_lines = sorted(cls_lines, key=lambda x: x[0])
    return [';'.join(i) for i in cls_lines], loc
