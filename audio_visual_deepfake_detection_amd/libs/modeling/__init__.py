from .blocks import (MaskedConv1D, MaskedMHCA, LayerNorm, TransformerBlock, Scale, AffineDropPath,
                     DeepInterpolator, LocalMaskedMHCA, MaskedMMHCA, LocalMaskedMMHCA,
                     MutilModelTransformerBlock)
from .models import make_backbone, make_neck, make_meta_arch, make_generator
from . import backbones       # noqa: F401  (registers the backbone)
from . import necks           # noqa: F401
from . import loc_generators  # noqa: F401
from . import meta_archs      # noqa: F401

__all__ = ['MaskedConv1D', 'MaskedMHCA', 'LayerNorm', 'TransformerBlock', 'Scale', 'AffineDropPath',
           'DeepInterpolator', 'make_backbone', 'make_neck', 'make_meta_arch', 'make_generator']
