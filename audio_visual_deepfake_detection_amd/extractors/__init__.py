"""Frozen feature extractors of stage 1 (the reference's audio_feature/ and video_feature/ scripts)."""
from .byola import AudioNTT2020Task6, LogMelFrontEnd, byola_features   # noqa: F401
from .emotion2vec import Emotion2Vec, emotion2vec_features   # noqa: F401
from .c3d import C3DVideoEncoder, get_video_encoder, video_features, chunk_plan   # noqa: F401
