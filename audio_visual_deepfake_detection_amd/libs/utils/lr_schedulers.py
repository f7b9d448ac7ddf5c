"""Per-iteration learning-rate schedules of the reference as closed-form functions of the iteration index.

The reference wraps torch optimisers in `_LRScheduler` subclasses (libs/utils/lr_schedulers.py:10-120, 123-212) built by
`make_scheduler` (libs/utils/train_utils.py:147-215) and steps them every iteration.  engine.TrainStep keeps the
learning rate in a device scalar next to the fused AdamW kernel, so here a schedule is just `lr(it)`: the value the
reference's `optimizer.param_groups[0]['lr']` holds while iteration `it` (0-based) runs.  The chainable recursions of
the reference equal these closed forms up to float rounding (pinned by tests/golden/lr_schedules.npz, rel. 1e-12)."""
import math
from bisect import bisect_right


class LinearWarmupCosineAnnealingLR:
    """lr_schedulers.py:10-120: linear warm-up over `warmup_epochs` iterations, then cosine to eta_min at max_epochs."""

    def __init__(self, base_lr, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=1e-8):
        self.base_lr, self.warmup_epochs, self.max_epochs = base_lr, warmup_epochs, max_epochs
        self.warmup_start_lr, self.eta_min = warmup_start_lr, eta_min

    def __call__(self, it):
        if it < self.warmup_epochs:
            return self.warmup_start_lr + it * (self.base_lr - self.warmup_start_lr) / (self.warmup_epochs - 1)
        return self.eta_min + 0.5 * (self.base_lr - self.eta_min) * (
            1 + math.cos(math.pi * (it - self.warmup_epochs) / (self.max_epochs - self.warmup_epochs)))


class LinearWarmupMultiStepLR:
    """lr_schedulers.py:123-212: linear warm-up, then x gamma at every milestone (counted after the warm-up)."""

    def __init__(self, base_lr, warmup_epochs, milestones, warmup_start_lr=0.0, gamma=0.1):
        self.base_lr, self.warmup_epochs, self.milestones = base_lr, warmup_epochs, sorted(milestones)
        self.warmup_start_lr, self.gamma = warmup_start_lr, gamma

    def __call__(self, it):
        if it < self.warmup_epochs:
            return self.warmup_start_lr + it * (self.base_lr - self.warmup_start_lr) / (self.warmup_epochs - 1)
        return self.base_lr * self.gamma ** bisect_right(self.milestones, it - self.warmup_epochs)


class CosineAnnealingLR:
    """torch.optim.lr_scheduler.CosineAnnealingLR(T_max, eta_min=0) -- the no-warm-up cosine branch (train_utils.py:193-199)."""

    def __init__(self, base_lr, t_max, eta_min=0.0):
        self.base_lr, self.t_max, self.eta_min = base_lr, t_max, eta_min

    def __call__(self, it):
        return self.eta_min + 0.5 * (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * it / self.t_max))


class MultiStepLR:
    def __init__(self, base_lr, milestones, gamma=0.1):
        self.base_lr, self.milestones, self.gamma = base_lr, sorted(milestones), gamma

    def __call__(self, it):
        return self.base_lr * self.gamma ** bisect_right(self.milestones, it)


def make_scheduler(optimizer_config, num_iters_per_epoch, base_lr=None):
    """Reference `make_scheduler` (train_utils.py:147-215) on cfg['opt']; returns `lr(it)`."""
    oc = optimizer_config
    base_lr = oc["learning_rate"] if base_lr is None else base_lr
    if oc.get("warmup", True):
        max_steps = (oc["epochs"] + oc["warmup_epochs"]) * num_iters_per_epoch
        warmup_steps = oc["warmup_epochs"] * num_iters_per_epoch
        if oc.get("schedule_type", "cosine") == "cosine":
            return LinearWarmupCosineAnnealingLR(base_lr, warmup_steps, max_steps)
        if oc["schedule_type"] == "multistep":
            return LinearWarmupMultiStepLR(base_lr, warmup_steps, [num_iters_per_epoch * s for s in oc["schedule_steps"]],
                                           gamma=oc.get("schedule_gamma", 0.1))
        raise TypeError("Unsupported scheduler!")
    max_steps = oc["epochs"] * num_iters_per_epoch
    if oc.get("schedule_type", "cosine") == "cosine":
        return CosineAnnealingLR(base_lr, max_steps)
    if oc["schedule_type"] == "multistep":
        return MultiStepLR(base_lr, [num_iters_per_epoch * s for s in oc["schedule_steps"]], oc.get("schedule_gamma", 0.1))
    raise TypeError("Unsupported scheduler!")
