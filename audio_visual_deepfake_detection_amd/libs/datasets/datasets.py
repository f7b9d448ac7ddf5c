"""Dataset plugin registry -- same names / call signatures as the reference's libs/datasets/datasets.py:5-43
(`make_dataset(cfg['dataset_name'], is_training, split, **cfg['dataset'])`, train.py:64)."""
import torch

from .data_utils import trivial_batch_collator, worker_init_reset_seed

datasets = {}


def register_dataset(name):
    def deco(cls):
        datasets[name] = cls
        return cls
    return deco


def make_dataset(name, is_training, split, **kwargs):
    return datasets[name](is_training, split, **kwargs)


def make_inference_dataset(name, is_training, split, sub_index, **kwargs):
    return datasets[name](is_training, split, sub_index, **kwargs)


def make_data_loader(dataset, is_training, generator, batch_size, num_workers):
    """Same loader policy as the reference (no auto-collation: the model batches the dicts itself).
    Items only carry host arrays -- all device work (K1 interpolation included) happens in the model's
    preprocessing, never in the forked workers."""
    return torch.utils.data.DataLoader(
        dataset, batch_size=batch_size, num_workers=num_workers, collate_fn=trivial_batch_collator,
        worker_init_fn=(worker_init_reset_seed if is_training else None), shuffle=is_training,
        drop_last=is_training, generator=generator, persistent_workers=num_workers > 0)
