"""Drop-in mirror of the reference's `libs` package API for the accelerated path
(registries, builders and class names of libs/core, libs/modeling, libs/datasets, libs/utils)."""
from . import core, modeling, datasets, utils   # noqa: F401
