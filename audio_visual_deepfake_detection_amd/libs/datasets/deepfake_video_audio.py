"""AV-Deepfake1M feature datasets (reference: libs/datasets/deepfake_video_audio.py).

Same registry names, constructor kwargs, on-disk inputs (`train_txt` of relative *.json paths, `json_folder`
metadata, per-modality `<rel>.npy` [T, C] fp32 features; inference: `test_folder/deepfake_test_sub{i}.txt`) and
the same per-item metadata as the reference classes.  One deliberate difference: `__getitem__` does NOT
interpolate on the CPU.  It returns the trimmed raw streams (`streams`: list of [T_s, C_s] arrays) plus the
random crop window (`crop`), and the meta-arch's preprocessing runs the K1 interpolation+concat kernel on the
device and applies the crop there (the reference spends 65 ms/clip of CPU time on this,
deepfake_video_audio.py:1722-1756).  No HIP call happens in a DataLoader worker.

Items are `ClipItem` dicts: every key of the reference's item dict (deepfake_video_audio.py:1820-1830) is there, and
`item['feats']` ([C, T] fp32, the reference's interpolated + concatenated + cropped matrix) is produced on first
access for consumers written against the reference dict -- in the loader process, where the reference computes it.
This package's own model never reads it (it takes `streams` + `crop`).  `return_feats=True` materialises it eagerly.
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .datasets import register_dataset
from .data_utils import truncate_window

BYOLA_FPS, EMOTION_FPS = 12.497, 50            # deepfake_video_audio.py:1547-1548


def interp_concat_host(streams, out_len, crop=None):
    """The interpolation block of the reference's `__getitem__` (deepfake_video_audio.py:1722-1756, :1017-1040 for the
    two-stream exp5 class) for readers of `item['feats']`: every stream transposed to [C_s, T_s], linearly resampled to
    `out_len` (align_corners=False; the video stream only when its length differs), concatenated on channels, then the
    crop window of truncate_feats (data_utils.py:124).  Host-side loader work, as in the reference; the model path runs
    the K1 kernel on the device instead."""
    import torch.nn.functional as F
    outs = []
    for i, arr in enumerate(streams):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(arr, np.float32).transpose()))
        if i > 0 or t.shape[-1] != out_len:
            t = F.interpolate(t.unsqueeze(0), size=out_len, mode='linear', align_corners=False).squeeze(0)
        outs.append(t)
    feats = torch.cat(outs, dim=0)
    if crop is not None:
        feats = feats[:, crop[0]:crop[1]].clone()
    return feats


class ClipItem(dict):
    """Item dict with a lazily computed `feats` entry (see the module docstring).  `max_seq_len` is an attribute, not a
    key: the keys are exactly the reference's plus `streams` / `crop` (instance attributes of a dict subclass pickle with
    it, so items survive the DataLoader workers' queues)."""

    max_seq_len = 768

    def __missing__(self, key):
        if key != 'feats':
            raise KeyError(key)
        feats = interp_concat_host(self['streams'], self.max_seq_len, self.get('crop'))
        self['feats'] = feats
        return feats


def _av_labels(name):
    """video/audio realness flags from the metadata file name (deepfake_video_audio.py:1566-1577)."""
    for key, val in (('fake_video_real_audio.json', [0, 1]), ('fake_video_fake_audio.json', [0, 0]),
                     ('real_video_fake_audio.json', [1, 0]), ('real.json', [1, 1])):
        if key in name:
            return val
    return [-1, -1]


class _TrainBase(Dataset):
    """Shared logic of the three training datasets.  Subclasses define `_load_streams`."""

    frame_labels = False

    def _setup(self, is_training, split, train_txt, json_folder, feat_stride, num_frames, default_fps, downsample_rate,
               max_seq_len, trunc_thresh, crop_ratio, video_input_dim, audio_input_dim, num_classes, file_prefix,
               file_ext, audio_file_ext, force_upsampling):
        assert crop_ratio is None or len(crop_ratio) == 2
        assert num_classes == 1
        self.is_training, self.split = is_training, split
        self.train_txt, self.json_folder = train_txt, json_folder
        self.feat_stride, self.num_frames, self.default_fps = feat_stride, num_frames, None
        self.downsample_rate, self.max_seq_len, self.trunc_thresh = downsample_rate, max_seq_len, trunc_thresh
        self.crop_ratio, self.video_input_dim, self.audio_input_dim = crop_ratio, video_input_dim, audio_input_dim
        self.num_classes, self.file_prefix = num_classes, file_prefix or ''
        self.file_ext, self.audio_file_ext, self.force_upsampling = file_ext, audio_file_ext, force_upsampling
        self.label_dict = {'Fake': 0}
        self.return_feats = getattr(self, 'return_feats', False)
        with open(train_txt, 'r') as f:
            self.data_list = [line.strip() for line in f.readlines()]
        self.db_attributes = {'dataset_name': 'DeepFake_Audio', 'tiou_thresholds': np.linspace(0.5, 0.95, 10),
                              'empty_label_ids': []}
        assert feat_stride > 0 and force_upsampling, \
            "the shipped configs all use force_upsampling with feat_stride > 0 (deepfake_video_audio.py:1699-1704)"

    def get_attributes(self):
        return self.db_attributes

    def __len__(self):
        return len(self.data_list)

    def _load_json_db(self, json_file):
        with open(os.path.join(self.json_folder, json_file), 'r') as fid:
            value = json.load(fid)
        duration = value['audio_frames'] / 16000
        if 'fps' in value:
            fps = value['fps']
        elif 'video_frames' in value:
            fps = value['video_frames'] / duration
        else:
            assert False, "Unknown video FPS."
        segments = labels = None
        if ('fake_segments' in value) and (len(value['fake_segments']) > 0):
            segments = np.asarray(value['fake_segments'], dtype=np.float32).reshape(-1, 2)
            labels = np.zeros([segments.shape[0]], dtype=np.int64)
        return {'fps': fps, 'duration': duration, 'split': value['split'].lower(), 'segments': segments,
                'labels': labels}

    def __getitem__(self, idx):
        rel = self.data_list[idx]
        item = self._load_json_db(rel)
        streams = self._load_streams(rel.replace('.json', '.npy'), item['duration'])
        n_video = streams[0].shape[0]
        feat_stride = float((n_video - 1) * self.feat_stride + self.num_frames) / self.max_seq_len
        num_frames = feat_stride
        feat_offset = 0.5 * num_frames / feat_stride
        feat_len = self.max_seq_len
        segments = labels = None
        if item['segments'] is not None:
            segments = torch.from_numpy(item['segments'] * item['fps'] / feat_stride - feat_offset)
            labels = torch.from_numpy(item['labels'])
            if self.is_training:
                vid_len = feat_len + feat_offset
                keep_s, keep_l = [], []
                for seg, lab in zip(segments, labels):
                    if seg[0] >= vid_len:
                        continue
                    ratio = (min(seg[1].item(), vid_len) - seg[0].item()) / (seg[1].item() - seg[0].item())
                    if ratio >= self.trunc_thresh:
                        keep_s.append(seg.clamp(max=vid_len))
                        keep_l.append(lab.view(1))
                if not keep_s:
                    raise RuntimeError(f"no valid segment left for {rel}")
                segments, labels = torch.stack(keep_s, dim=0), torch.cat(keep_l)
        crop = None
        if self.is_training and segments is not None:
            st, ed, segments, labels = truncate_window(feat_len, segments, labels, self.max_seq_len, self.trunc_thresh,
                                                       feat_offset, self.crop_ratio)
            crop = (st, ed)
        out = ClipItem({'video_id': rel.replace('.json', '.mp4'), 'streams': streams, 'crop': crop, 'segments': segments,
                        'n_fakes': 0 if segments is None else segments.shape[0], 'labels': labels,
                        'av_labels': torch.tensor(_av_labels(rel)), 'fps': item['fps'], 'duration': item['duration'],
                        'feat_stride': feat_stride, 'split': item['split'], 'feat_num_frames': num_frames})
        out.max_seq_len = self.max_seq_len
        if self.return_feats:
            out['feats']
        if self.frame_labels:
            out['gt_frame_labels'] = None
            if self.is_training and segments is not None:
                out['gt_frame_labels'] = self.map_segments_to_labels(out)
        return out

    def map_segments_to_labels(self, d):
        """Frame-level targets of exp13 (deepfake_video_audio.py:2012-2033; units mixed exactly as there)."""
        lab = torch.zeros(768)
        seg_len = d['duration'] / 768
        for start, end in d['segments']:
            if d['av_labels'][0] > 0 or d['av_labels'][1] > 0:
                lab[int(start / seg_len):int(end / seg_len)] = 1
        return lab


@register_dataset("deepfake_video_audioEmoBYOLA")
class DeepFakeVideoAudioDataset(_TrainBase):
    """exp10: LAV-DF video + BYOL-A + Emotion2Vec (deepfake_video_audio.py:1467-1839)."""

    def __init__(self, is_training, split, video_feat_folder, audio_feat_folder=None, audio_byola_feat_folder=None,
                 audio_emo_feat_folder=None, train_txt=None, json_folder=None, feat_stride=1, num_frames=1,
                 default_fps=None, downsample_rate=0, max_seq_len=768, trunc_thresh=0.5, crop_ratio=None,
                 video_input_dim=256, audio_input_dim=2816, num_classes=1, file_prefix=None, file_ext='.npy',
                 audio_file_ext='.npy', force_upsampling=True, return_feats=False):
        self.video_feat_folder = video_feat_folder
        self.return_feats = return_feats
        self.audio_byola_feat_folder, self.audio_emo_feat_folder = audio_byola_feat_folder, audio_emo_feat_folder
        self._setup(is_training, split, train_txt, json_folder, feat_stride, num_frames, default_fps, downsample_rate,
                    max_seq_len, trunc_thresh, crop_ratio, video_input_dim, audio_input_dim, num_classes, file_prefix,
                    file_ext, audio_file_ext, force_upsampling)
        self.byola_fps, self.emotion_fps = BYOLA_FPS, EMOTION_FPS

    def _load_streams(self, rel_npy, duration):
        v = np.load(os.path.join(self.video_feat_folder, rel_npy))
        a1 = np.load(os.path.join(self.audio_byola_feat_folder, rel_npy))
        a2 = np.load(os.path.join(self.audio_emo_feat_folder, rel_npy))
        a1 = a1[:int(self.byola_fps * duration - 0.3657)]          # deepfake_video_audio.py:1683-1684
        a2 = a2[:int(self.emotion_fps * duration - 0.817)]
        return [np.ascontiguousarray(v, np.float32), np.ascontiguousarray(a1, np.float32),
                np.ascontiguousarray(a2, np.float32)]


@register_dataset("deepfake_video_audioEmoBYOLA_THE")
class DeepFakeVideoAudioDatasetTHE(DeepFakeVideoAudioDataset):
    """exp13: as exp10 plus frame-level labels (deepfake_video_audio.py:1841-2237)."""
    frame_labels = True


@register_dataset("deepfake_video_audio")
class DeepFakeVideoEmoDataset(_TrainBase):
    """exp5: LAV-DF video + Emotion2Vec only (deepfake_video_audio.py:739-1102)."""

    def __init__(self, is_training, split, video_feat_folder, audio_feat_folder, train_txt, json_folder, feat_stride,
                 num_frames, default_fps, downsample_rate, max_seq_len, trunc_thresh, crop_ratio, video_input_dim,
                 audio_input_dim, num_classes, file_prefix, file_ext, audio_file_ext, force_upsampling, return_feats=False):
        self.video_feat_folder, self.audio_feat_folder = video_feat_folder, audio_feat_folder
        self.return_feats = return_feats
        self._setup(is_training, split, train_txt, json_folder, feat_stride, num_frames, default_fps, downsample_rate,
                    max_seq_len, trunc_thresh, crop_ratio, video_input_dim, audio_input_dim, num_classes, file_prefix,
                    file_ext, audio_file_ext, force_upsampling)
        self.emotion_fps = EMOTION_FPS

    def _load_streams(self, rel_npy, duration):
        v = np.load(os.path.join(self.video_feat_folder, rel_npy))
        a = np.load(os.path.join(self.audio_feat_folder, rel_npy))
        a = a[:int(self.emotion_fps * duration - 0.817)]           # deepfake_video_audio.py:955
        return [np.ascontiguousarray(v, np.float32), np.ascontiguousarray(a, np.float32)]


@register_dataset("deepfake_video_audioEmoBYOLA_inference")
class DeepFakeVideoAudioDatasetInfer3(Dataset):
    """exp12 test-time dataset (deepfake_video_audio.py:351-558): ids and durations from
    `test_folder/deepfake_test_sub{sub_index}.txt`, fps = n_video_rows / duration."""

    def __init__(self, is_training, split, sub_index, crop_ratio=None, default_fps=None, downsample_rate=0,
                 video_feat_folder=None, audio_feat_folder=None, audio_byola_feat_folder=None,
                 audio_emo_feat_folder=None, audio_file_ext='.npy', num_classes=1, input_dim=0, video_input_dim=256,
                 audio_input_dim=2816, feat_stride=1, num_frames=1, test_folder=None, trunc_thresh=0.5,
                 max_seq_len=768, force_upsampling=True, return_feats=False, **_unused):
        assert num_classes == 1 and feat_stride > 0 and force_upsampling
        self.sub_index, self.test_folder, self.return_feats = sub_index, test_folder, return_feats
        self.video_feat_folder = video_feat_folder
        self.audio_byola_feat_folder, self.audio_emo_feat_folder = audio_byola_feat_folder, audio_emo_feat_folder
        self.feat_stride, self.num_frames, self.max_seq_len = feat_stride, num_frames, max_seq_len
        self.byola_fps, self.emotion_fps = BYOLA_FPS, EMOTION_FPS
        self.data_list = []
        with open(os.path.join(test_folder, f"deepfake_test_sub{sub_index}.txt"), 'r') as f:
            for line in f.readlines():
                items = line.strip().split(',')
                self.data_list.append({'id': items[0], 'duration': float(items[1])})
        self.db_attributes = {'dataset_name': 'DeepFake_Audio', 'tiou_thresholds': np.linspace(0.5, 0.95, 10),
                              'empty_label_ids': []}

    def get_attributes(self):
        return self.db_attributes

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, idx):
        it = self.data_list[idx]
        rel = it['id'].replace('.mp4', '.npy')
        v = np.load(os.path.join(self.video_feat_folder, rel))
        a1 = np.load(os.path.join(self.audio_byola_feat_folder, rel))[:int(self.byola_fps * it['duration'] - 0.3657)]
        a2 = np.load(os.path.join(self.audio_emo_feat_folder, rel))[:int(self.emotion_fps * it['duration'] - 0.817)]
        feat_stride = float((v.shape[0] - 1) * self.feat_stride + self.num_frames) / self.max_seq_len
        out = ClipItem({'video_id': it['id'],
                        'streams': [np.ascontiguousarray(x, np.float32) for x in (v, a1, a2)], 'crop': None,
                        'fps': v.shape[0] / it['duration'], 'duration': it['duration'], 'feat_stride': feat_stride,
                        'feat_num_frames': feat_stride})
        out.max_seq_len = self.max_seq_len
        if self.return_feats:
            out['feats']
        return out
