"""TEST INFRASTRUCTURE ONLY -- a tiny synthetic AV-Deepfake1M-style tree (metadata json + per-modality .npy
features + list files), regenerated from a seed so fixtures need not store it."""
import json
import os

import numpy as np

KINDS = ["real.json", "fake_video_fake_audio.json", "real_video_fake_audio.json", "fake_video_real_audio.json"]


def make_synth_tree(root, seed=0, n=6, dims=(8, 16, 8)):
    rs = np.random.RandomState(seed)
    for d in ("meta", "video", "byola", "emo", "test_folder"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rels, test_lines = [], []
    for i in range(n):
        kind = KINDS[i % 4]
        rel = f"id{i:03d}/clip/{kind}"
        n_video = int(rs.randint(120, 400)) if i != 3 else 768          # one clip already at max_seq_len
        duration = n_video / 25.0
        segs = []
        if kind != "real.json":
            k = 1 + i % 2
            st = np.sort(rs.uniform(0.2, duration - 1.5, size=k))
            segs = [[float(a), float(a + rs.uniform(0.2, 1.2))] for a in st]
        meta = {"audio_frames": int(round(duration * 16000)), "video_frames": n_video, "fake_segments": segs,
                "split": "train"}
        os.makedirs(os.path.dirname(os.path.join(root, "meta", rel)), exist_ok=True)
        with open(os.path.join(root, "meta", rel), "w") as f:
            json.dump(meta, f)
        dur = meta["audio_frames"] / 16000
        lens = (n_video, int(12.497 * dur - 0.3657) + 3, int(50 * dur - 0.817) + 2)     # audio saved un-trimmed
        for sub, t, c in zip(("video", "byola", "emo"), lens, dims):
            p = os.path.join(root, sub, rel.replace(".json", ".npy"))
            os.makedirs(os.path.dirname(p), exist_ok=True)
            np.save(p, rs.standard_normal((t, c)).astype(np.float32))
        rels.append(rel)
        test_lines.append(f"{rel.replace('.json', '.mp4')},{dur}")
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join(rels) + "\n")
    with open(os.path.join(root, "test_folder", "deepfake_test_sub1.txt"), "w") as f:
        f.write("\n".join(test_lines) + "\n")
    return rels


def dataset_kwargs(root, dims=(8, 16, 8)):
    return dict(video_feat_folder=os.path.join(root, "video"), audio_feat_folder=None,
                audio_byola_feat_folder=os.path.join(root, "byola"), audio_emo_feat_folder=os.path.join(root, "emo"),
                train_txt=os.path.join(root, "train.txt"), json_folder=os.path.join(root, "meta"), feat_stride=1,
                num_frames=1, default_fps=None, downsample_rate=0, max_seq_len=768, trunc_thresh=0.5,
                crop_ratio=[0.9, 1.0], video_input_dim=dims[0], audio_input_dim=dims[1] + dims[2], num_classes=1,
                file_prefix="rgb", file_ext=".npy", audio_file_ext=".npy", force_upsampling=True)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 1: the reference's own `tinydataset/metadata/**.json` (12 AV-Deepfake1M metadata files: DATA) as a
# compact fixture, plus seeded synthetic features of the shapes the reference's extractors would write for them
# ---------------------------------------------------------------------------------------------------------------
TINY_FIELDS = ("file", "split", "fake_segments", "video_frames", "audio_frames")


def tiny_tree_from_fixture(root, fixture_json, seed=0, dims=(256, 768)):
    """Rebuild the tinydataset tree under `root` from tests/golden/tinydataset_meta.json: metadata json per clip (the
    fields the dataset classes read), `train.txt`, and per clip a LAV-DF-like video feature file [video_frames, 256]
    and an Emotion2Vec-like audio feature file [(audio_frames - 400)//320 + 1, 768] (the lengths those encoders emit:
    SURVEY.md section 8b), N(0,1) from RandomState(seed + index)."""
    with open(fixture_json) as f:
        metas = json.load(f)
    rels = []
    for i, m in enumerate(metas):
        rel = m["file"].replace(".mp4", ".json")
        p = os.path.join(root, "meta", rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            json.dump({k: m[k] for k in TINY_FIELDS}, f)
        rs = np.random.RandomState(seed + i)
        tv, ta = int(m["video_frames"]), (int(m["audio_frames"]) - 400) // 320 + 1
        for sub, t, c in (("video", tv, dims[0]), ("emo", ta, dims[1])):
            q = os.path.join(root, sub, rel.replace(".json", ".npy"))
            os.makedirs(os.path.dirname(q), exist_ok=True)
            np.save(q, rs.standard_normal((t, c)).astype(np.float32))
        rels.append(rel)
    with open(os.path.join(root, "train.txt"), "w") as f:
        f.write("\n".join(rels) + "\n")
    return rels


def tiny_dataset_kwargs(root, dims=(256, 768)):
    """Constructor kwargs of the `deepfake_video_audio` dataset as configs_train/deepfake_exp5.yaml passes them."""
    return dict(video_feat_folder=os.path.join(root, "video"), audio_feat_folder=os.path.join(root, "emo"),
                train_txt=os.path.join(root, "train.txt"), json_folder=os.path.join(root, "meta"), feat_stride=1,
                num_frames=1, default_fps=None, downsample_rate=0, max_seq_len=768, trunc_thresh=0.5,
                crop_ratio=[0.9, 1.0], video_input_dim=dims[0], audio_input_dim=dims[1], num_classes=1,
                file_prefix="rgb", file_ext=".npy", audio_file_ext=".npy", force_upsampling=True)
