"""AP at the fixed tIoU thresholds {0.5, 0.75, 0.9, 0.95} -- the "AP@0.5" of the headline metric is `mAP[0]`.

Same call signature and arithmetic as the reference's libs/utils/Evaluation/eval_detection.py (`ANETdetection` :18-223,
`compute_average_precision_detection` :228-319): ground truth is the LIST of validation items the reference collects
(`video_id`, `n_fakes`, `segments_time`; :78-92), predictions are the json written by `post_process_multi`; the tIoU
thresholds are fixed whatever the caller passes (:36,247).  numpy only: no pandas, no `np.float` (the reference's own
module raises on numpy >= 1.24, :310-311) and no per-false-positive debugging print (:296).  Pinned to the reference's
output by tests/golden/eval_detection.npz."""
import json

import numpy as np

from ..metrics import compute_average_precision_detection as _ap_generic

TIOU = np.array([0.5, 0.75, 0.9, 0.95])


def compute_average_precision_detection(ground_truth, prediction, tiou_thresholds=TIOU):
    """eval_detection.py:228-319 -- thresholds are overwritten with the fixed set, as there."""
    return _ap_generic(ground_truth, prediction, TIOU)


class ANETdetection(object):
    GROUND_TRUTH_FIELDS = ['database']
    PREDICTION_FIELDS = ['results', 'version', 'external_data']

    def __init__(self, ground_truth_filename=None, prediction_filename=None, ground_truth_fields=GROUND_TRUTH_FIELDS,
                 prediction_fields=PREDICTION_FIELDS, dataset_name='', tiou_thresholds=TIOU, subset='validation',
                 verbose=False, check_status=False):
        if not ground_truth_filename:
            raise IOError('Please input a valid ground truth file.')
        if not prediction_filename:
            raise IOError('Please input a valid prediction file.')
        assert not check_status, "the blocked-video list of the ActivityNet server is not used by the deepfake configs"
        self.subset, self.verbose, self.dataset_name = subset, verbose, dataset_name
        self.tiou_thresholds = TIOU.copy()
        self.pred_fields = prediction_fields
        self.ap = None
        self.activity_index = {'Fake': 0}
        self.ground_truth = self._import_ground_truth(ground_truth_filename)
        self.prediction = self._import_prediction(prediction_filename)
        if self.verbose:
            print('[INIT] Loaded annotations from {} subset.'.format(subset))
            print('\tNumber of ground truth instances: {}'.format(len(self.ground_truth['label'])))
            print('\tNumber of predictions: {}'.format(len(self.prediction['label'])))
            print('\tFixed threshold for tiou score: {}'.format(self.tiou_thresholds))

    def _import_ground_truth(self, items):
        vid, ts, te = [], [], []
        for v in items:
            if v['n_fakes'] == 0:
                continue
            for ann in v['segments_time']:
                vid.append(v['video_id'].strip())
                ts.append(float(ann[0]))
                te.append(float(ann[1]))
        return {'video-id': np.asarray(vid, dtype=object), 't-start': np.asarray(ts, np.float64),
                't-end': np.asarray(te, np.float64), 'label': np.zeros(len(vid), np.int64)}

    def _import_prediction(self, prediction_filename):
        with open(prediction_filename, 'r') as fobj:
            data = json.load(fobj)
        if not all(field in data.keys() for field in self.pred_fields):
            raise IOError('Please input a valid prediction file.')
        vid, ts, te, lab, sc = [], [], [], [], []
        for videoid, v in data['results'].items():
            for result in v:
                vid.append(videoid)
                ts.append(float(result['segment'][0]))
                te.append(float(result['segment'][1]))
                lab.append(self.activity_index[result['label']])
                sc.append(result['score'])
        return {'video-id': np.asarray(vid, dtype=object), 't-start': np.asarray(ts, np.float64),
                't-end': np.asarray(te, np.float64), 'label': np.asarray(lab, np.int64), 'score': np.asarray(sc, np.float64)}

    def wrapper_compute_average_precision(self):
        ap = np.zeros((len(self.tiou_thresholds), len(self.activity_index)))
        assert len(self.ground_truth['label']) > 0, "no ground-truth segment (the reference's groupby raises here too)"
        ap[:, 0] = compute_average_precision_detection(self.ground_truth, self.prediction)
        return ap

    def evaluate(self):
        self.ap = self.wrapper_compute_average_precision()
        self.mAP = self.ap.mean(axis=1)
        self.average_mAP = self.mAP.mean()
        if self.verbose:
            print(f'[RESULTS] Performance on {self.dataset_name} detection task.')
            print('Average-mAP: {}'.format(self.average_mAP))
        return self.mAP, self.average_mAP
