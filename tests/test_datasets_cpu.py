"""Host logic of the dataset plugins (no GPU): file listing, trimming, feat_stride / segment-grid maths, the seeded
random crop of truncate_feats and the exp13 frame labels must equal what the REFERENCE dataset classes produced on the
same synthetic tree (tests/golden/datasets.npz, written by oracle/make_golden.py).  The feature matrix itself is
checked by resampling this package's raw `streams` with the CPU oracle and comparing checksums with the reference's
`feats` (the product does that step on the device: tests/test_gpu_model.py::test_dataset_items_through_model)."""
import os
import random

import numpy as np
import torch

from oracle import ummaformer_ref as R
from oracle.synth_data import make_synth_tree, dataset_kwargs


def _load_datasets_module():
    """libs/datasets has no HIP dependency; import it without triggering the package-level .so load."""
    import importlib.util
    import sys
    import types
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "audio_visual_deepfake_detection_amd", "libs", "datasets")
    pkg = types.ModuleType("avdf_datasets")
    pkg.__path__ = [root]
    sys.modules["avdf_datasets"] = pkg
    for name in ("data_utils", "datasets", "deepfake_video_audio"):
        spec = importlib.util.spec_from_file_location(f"avdf_datasets.{name}", os.path.join(root, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"avdf_datasets.{name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["avdf_datasets.datasets"]


def _feats(item):
    f = R.interp_concat(item["streams"], 768)
    st, ed = item["crop"] if item["crop"] is not None else (0, 768)
    return f[:, st:ed]


def test_training_and_inference_items_match_reference(tmp_path, golden_dir):
    g = np.load(os.path.join(golden_dir, "datasets.npz"))
    D = _load_datasets_module()
    root = str(tmp_path)
    make_synth_tree(root, seed=0)
    kw = dataset_kwargs(root)
    for tag, name in (("a", "deepfake_video_audioEmoBYOLA"), ("t", "deepfake_video_audioEmoBYOLA_THE")):
        ds = D.make_dataset(name, True, ["train"], **kw)
        assert len(ds) == int(g["n"])
        assert set(ds.get_attributes()) == {"dataset_name", "tiou_thresholds", "empty_label_ids"}
        for i in range(len(ds)):
            random.seed(100 + i)
            d = ds[i]
            f = _feats(d)
            assert tuple(f.shape) == tuple(g[f"{tag}{i}_shape"])
            assert np.allclose([float(f.double().sum()), float(f.double().abs().sum())], g[f"{tag}{i}_sum"], rtol=1e-9)
            assert np.array_equal(f[:, 0].numpy(), g[f"{tag}{i}_col0"]) and np.array_equal(f[:, -1].numpy(), g[f"{tag}{i}_colN"])
            seg = np.zeros((0, 2), np.float32) if d["segments"] is None else d["segments"].numpy()
            assert np.array_equal(seg, g[f"{tag}{i}_segments"]), (i, seg, g[f"{tag}{i}_segments"])
            meta = np.array([d["fps"], d["duration"], d["feat_stride"], d["feat_num_frames"]], np.float64)
            assert np.array_equal(meta, g[f"{tag}{i}_meta"])
            assert np.array_equal(d["av_labels"].numpy(), g[f"{tag}{i}_av"])
            if tag == "t":
                gl = np.zeros(0, np.float32) if d["gt_frame_labels"] is None else d["gt_frame_labels"].numpy()
                assert np.array_equal(gl, g[f"{tag}{i}_frames"])
    dsi = D.make_inference_dataset(
        "deepfake_video_audioEmoBYOLA_inference", False, ["test"], 1, crop_ratio=None, default_fps=None,
        downsample_rate=0, video_feat_folder=kw["video_feat_folder"], audio_feat_folder=None,
        audio_byola_feat_folder=kw["audio_byola_feat_folder"], audio_emo_feat_folder=kw["audio_emo_feat_folder"],
        audio_file_ext=".npy", num_classes=1, input_dim=0, video_input_dim=8, audio_input_dim=24, feat_stride=1,
        num_frames=1, test_folder=os.path.join(root, "test_folder"), trunc_thresh=0.5, max_seq_len=768,
        force_upsampling=True)
    for i in range(len(dsi)):
        d = dsi[i]
        f = _feats(d)
        assert np.allclose([float(f.double().sum()), float(f.double().abs().sum())], g[f"i{i}_sum"], rtol=1e-9)
        meta = np.array([d["fps"], d["duration"], d["feat_stride"], d["feat_num_frames"]], np.float64)
        assert np.array_equal(meta, g[f"i{i}_meta"])


def test_loader_yields_lists_of_dicts(tmp_path):
    D = _load_datasets_module()
    root = str(tmp_path)
    make_synth_tree(root, seed=1, n=4)
    ds = D.make_dataset("deepfake_video_audioEmoBYOLA", True, ["train"], **dataset_kwargs(root))
    loader = D.make_data_loader(ds, True, torch.Generator().manual_seed(0), batch_size=2, num_workers=0)
    batch = next(iter(loader))
    assert isinstance(batch, list) and len(batch) == 2 and "streams" in batch[0]


def test_tinydataset_metadata_items_match_reference(tmp_path, golden_dir):
    """The reference's own 12 `tinydataset/metadata/**.json` files (tests/golden/tinydataset_meta.json keeps the fields the
    dataset classes read) through the `deepfake_video_audio` (exp5) dataset plugin: metadata, grid-unit segments, the
    seeded training crop and the resampled feature matrix equal the reference class's items (libs/datasets/
    deepfake_video_audio.py:739-1102) on the same seeded feature files."""
    from oracle.synth_data import tiny_tree_from_fixture, tiny_dataset_kwargs
    g = np.load(os.path.join(golden_dir, "tinydataset.npz"), allow_pickle=True)
    D = _load_datasets_module()
    root = str(tmp_path)
    tiny_tree_from_fixture(root, os.path.join(golden_dir, "tinydataset_meta.json"), seed=500)
    kw = tiny_dataset_kwargs(root)
    ds_eval = D.make_dataset("deepfake_video_audio", False, ["train"], **kw)
    ds_train = D.make_dataset("deepfake_video_audio", True, ["train"], **kw)
    assert len(ds_eval) == int(g["n"]) == 12
    for i in range(12):
        d = ds_eval[i]
        f = R.interp_concat(d["streams"], 768)
        assert tuple(f.shape) == (1024, 768)
        assert np.allclose([float(f.double().sum()), float(f.double().abs().sum())], g[f"e{i}_sum"], rtol=1e-9)
        seg = np.zeros((0, 2), np.float32) if d["segments"] is None else d["segments"].numpy()
        assert np.array_equal(seg, g[f"e{i}_segments"])
        assert np.array_equal(np.array([d["fps"], d["duration"], d["feat_stride"], d["feat_num_frames"]], np.float64), g[f"e{i}_meta"])
        assert np.array_equal(d["av_labels"].numpy(), g[f"e{i}_av"])
        random.seed(900 + i)
        t = ds_train[i]
        ft = _feats(t)
        assert tuple(ft.shape) == tuple(g[f"t{i}_shape"])
        assert np.allclose([float(ft.double().sum()), float(ft.double().abs().sum())], g[f"t{i}_sum"], rtol=1e-9)
        segt = np.zeros((0, 2), np.float32) if t["segments"] is None else t["segments"].numpy()
        assert np.array_equal(segt, g[f"t{i}_segments"])
