"""MI355X-native (gfx950) hot path of the audio-visual deepfake temporal-localisation system.

    from audio_visual_deepfake_detection_amd import libs          # drop-in for the reference's `libs`
    model = libs.modeling.make_meta_arch(cfg['model_name'], **cfg['model']).cuda()

Importing the package loads libavdf_hip.so (built by csrc/build.py); there is no CPU fallback.
"""
import os

from . import _lib          # noqa: F401  loud failure if the HIP extension is missing
from . import ops           # noqa: F401
from . import libs          # noqa: F401

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


def build_model(yaml_name="deepfake_exp10.yaml", model_name=None, **model_overrides):
    """Convenience: meta-arch from a packaged YAML with optional cfg['model'] overrides (and meta-arch name)."""
    from .libs.core import load_config
    from .libs.modeling import make_meta_arch
    cfg = load_config(os.path.join(CONFIG_DIR, yaml_name))
    if model_name is not None:
        cfg["model_name"] = model_name
    ds_keys = ("video_input_dim", "audio_input_dim", "max_seq_len", "num_classes")
    for k, v in model_overrides.items():
        cfg["model"][k] = v
        if k in ds_keys:
            cfg["dataset"][k] = v
    return make_meta_arch(cfg["model_name"], **cfg["model"]), cfg
