from .data_utils import worker_init_reset_seed, truncate_window
from .datasets import make_dataset, make_data_loader, make_inference_dataset, register_dataset
from . import deepfake_video_audio  # noqa: F401  (registers the datasets)

__all__ = ['worker_init_reset_seed', 'truncate_window', 'make_dataset', 'make_data_loader',
           'make_inference_dataset', 'register_dataset']
