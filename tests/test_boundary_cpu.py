"""Drop-in boundary obligations that need no GPU: optimiser parameter partition, checkpoint key
layout incl. the `module.` prefix of the reference's DataParallel checkpoints, dataset item dict."""
import hashlib
import os
import random

import numpy as np
import pytest
import torch

from audio_visual_deepfake_detection_amd import build_model
from audio_visual_deepfake_detection_amd.engine import split_decay


@pytest.mark.parametrize("tag,yaml_name", [("exp10", "deepfake_exp10.yaml"), ("exp5", "deepfake_exp5.yaml"),
                                           ("exp13", "deepfake_exp13.yaml")])
def test_split_decay_equals_reference_make_optimizer(golden_dir, tag, yaml_name):
    """engine.split_decay == the decay / no-decay groups the reference's make_optimizer builds for its own model
    (libs/utils/train_utils.py:65-122; exp10: 195 / 413): same counts, same sorted name lists (sha256)."""
    g = np.load(os.path.join(golden_dir, "optimizer_split.npz"))
    model, cfg = build_model(yaml_name)
    decay, no_decay = split_decay(model)
    assert (len(decay), len(no_decay)) == tuple(int(v) for v in g[f"{tag}_counts"])
    assert hashlib.sha256("\n".join(sorted(decay)).encode()).hexdigest() == str(g[f"{tag}_decay_sha"])
    assert hashlib.sha256("\n".join(sorted(no_decay)).encode()).hexdigest() == str(g[f"{tag}_nodecay_sha"])
    assert not set(decay) & set(no_decay)
    assert set(decay) | set(no_decay) == {n for n, _ in model.named_parameters()}
    assert cfg["opt"]["weight_decay"] == float(g[f"{tag}_hparams"][0]) and cfg["opt"]["learning_rate"] == float(g[f"{tag}_hparams"][1])


def test_module_prefixed_ema_checkpoint_loads():
    """inference.py:63-76 wraps the model in nn.DataParallel and loads checkpoint['state_dict_ema'], whose keys carry the
    `module.` prefix: the same call sequence must work on this package's model (CPU: no forward involved)."""
    from oracle import ummaformer_ref as R
    small = dict(video_input_dim=8, audio_input_dim=88, max_seq_len=192, embd_dim=32, fpn_dim=32, head_dim=32, n_head=4)
    model, _ = build_model("deepfake_exp10.yaml", **small)
    cfg = R.make_cfg(input_dim=96, max_seq_len=192, embd_dim=32, n_head=4, fpn_dim=32, head_dim=32)
    sd = R.synth_state_dict(R.param_shapes(cfg), 17, True)
    ckpt = {"state_dict_ema": {"module." + k: v for k, v in sd.items()}}
    wrapped = torch.nn.DataParallel(model, device_ids=None if not torch.cuda.is_available() else [0])
    missing = wrapped.load_state_dict(ckpt["state_dict_ema"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    got = model.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)


def test_dataset_item_has_reference_dict_keys_incl_feats(tmp_path):
    """`__getitem__` of the reference returns the dict of deepfake_video_audio.py:1820-1830 with `feats` [C, T]; this
    package's item carries the raw streams for the on-device K1 and produces `feats` lazily for consumers of the
    reference dict (CPU interpolation in the loader process, exactly where the reference computes it)."""
    from audio_visual_deepfake_detection_amd.libs.datasets import make_dataset
    from oracle import ummaformer_ref as R
    from oracle.synth_data import make_synth_tree, dataset_kwargs
    root = str(tmp_path)
    make_synth_tree(root, seed=0)
    ds = make_dataset("deepfake_video_audioEmoBYOLA", True, ["train"], **dataset_kwargs(root))
    for i in range(len(ds)):
        random.seed(100 + i)
        d = ds[i]
        for k in ("video_id", "segments", "n_fakes", "labels", "av_labels", "fps", "duration", "feat_stride", "split",
                  "feat_num_frames"):
            assert k in d, k
        f = d["feats"]                                              # lazily materialised
        ref = R.interp_concat(d["streams"], 768)
        st, ed = d["crop"] if d["crop"] is not None else (0, 768)
        assert f.dtype == torch.float32 and torch.equal(f, ref[:, st:ed])
        assert "feats" in d                                         # cached after the first access
    eager = make_dataset("deepfake_video_audioEmoBYOLA", False, ["train"], return_feats=True, **dataset_kwargs(root))
    assert "feats" in dict(eager[0]) and tuple(eager[0]["feats"].shape) == (32, 768)
