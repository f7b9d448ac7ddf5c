"""Frozen feature extractors of stage 1 (the reference's audio_feature/ and video_feature/ scripts)."""
from .byola import AudioNTT2020Task6, LogMelFrontEnd, byola_features   # noqa: F401
