"""Frozen feature extractors of stage 1 (the reference's audio_feature/ and video_feature/ scripts)."""
from .byola import AudioNTT2020Task6, LogMelFrontEnd, byola_features   # noqa: F401
from .emotion2vec import Emotion2Vec, emotion2vec_features   # noqa: F401
