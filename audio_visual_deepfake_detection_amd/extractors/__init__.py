"""Frozen feature extractors of stage 1 (the reference's audio_feature/ and video_feature/ scripts)."""
from .byola import AudioNTT2020Task6, LogMelFrontEnd, byola_features, byola_features_batch   # noqa: F401
from .emotion2vec import (Emotion2Vec, emotion2vec_features, emotion2vec_features_batch,   # noqa: F401
                          emotion2vec_features_equal_length)
from .c3d import C3DVideoEncoder, get_video_encoder, video_features, chunk_plan   # noqa: F401


def extract_clip_batch(frames, pcm, c3d, byola, front_end, e2v, outs, audio_stream=None):
    """Stage-1 features of a batch of clips written into preallocated device tensors.

    frames[b]: u8 [T,H,W,3]; pcm[b]: fp32 [N] @16 kHz; outs[b] = (video [Tv,256], byol-a [Ta,2048], emotion2vec [Te,768])
    destination tensors -- rows beyond a destination's length are dropped (the dataset's own trim,
    deepfake_video_audio.py:1683).  The three encoders are independent (extract_video_features.py,
    extract_audio_feature_batch.py x2 are separate scripts upstream): with `audio_stream` (a torch.cuda.Stream) the two
    audio encoders -- many small launches, 1535-row GEMMs -- run beside the C3D encoder's few chip-filling launches;
    results are identical, the caller's current stream waits for both."""
    import torch
    cur = torch.cuda.current_stream()
    if audio_stream is not None:
        audio_stream.wait_stream(cur)
    with torch.cuda.stream(audio_stream if audio_stream is not None else cur):
        # equal-length clips (a tensor [B, N]): one batched Emotion2Vec pass (row-wise GEMMs over B x T rows)
        fes = emotion2vec_features_equal_length(pcm, e2v) if torch.is_tensor(pcm) and pcm.dim() == 2 else None
        for b in range(len(outs)):
            fa = byola_features(pcm[b], byola, front_end)
            fe = fes[b] if fes is not None else emotion2vec_features(pcm[b], e2v)
            outs[b][1].copy_(fa[:outs[b][1].shape[0]])
            outs[b][2].copy_(fe[:outs[b][2].shape[0]])
    for b in range(len(outs)):
        fv = video_features(frames[b], c3d)
        outs[b][0].copy_(fv[:outs[b][0].shape[0]])
    if audio_stream is not None:
        cur.wait_stream(audio_stream)
