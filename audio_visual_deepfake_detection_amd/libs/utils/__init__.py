from .nms import batched_nms

__all__ = ['batched_nms']
