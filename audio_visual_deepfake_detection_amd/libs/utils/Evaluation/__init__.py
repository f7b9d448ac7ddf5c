"""Localisation evaluation of the deepfake configs (reference: libs/utils/Evaluation/): fixed tIoU {0.5, 0.75, 0.9, 0.95}
average precision on the segments dumped by validation / inference, plus the challenge submission formatting."""
from .eval import run_evaluation, post_process_multi, evaluation_detection, format_submission   # noqa: F401
from .eval_detection import ANETdetection, compute_average_precision_detection                   # noqa: F401
