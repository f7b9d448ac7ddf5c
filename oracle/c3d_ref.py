"""TEST INFRASTRUCTURE ONLY -- torch CPU restatement of the LAV-DF C3D visual feature path.

PARITY UNPINNED: `utils.Conv3d/Conv1d`, torchvision and alignvideo_model.ckpt are absent;
restates video_feature/aligned_video/model/video_encoder.py:27-62 with `utils.ConvNd := conv -> [BN] -> LeakyReLU`
and extract_video_features.py:19-77,136-166 (/255, tail pad to 512, nearest 96x96, chunk + stitch)."""
import torch
import torch.nn.functional as F

POOLS = {"block0.1": 3, "block1.1": 2, "block2.1": 2, "block3.0": 2, "block3.2": 2}
ORDER = ["block0.0", "block0.1", "block1.0", "block1.1", "block2.0", "block2.1", "block3.0", "block3.2"]


def _bn(sd, p, x):
    if p + ".bn.weight" in sd:
        x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                         sd[p + ".bn.bias"], False, 0.0, 1e-5)
    return x


def encoder_forward(sd, video):
    """video [B, 3, T, H, W] -> [B, 256, T]."""
    x = video
    for p in ORDER:
        x = F.leaky_relu(_bn(sd, p, F.conv3d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=1)), 0.01)
        if p in POOLS:
            x = F.max_pool3d(x, (1, POOLS[p], POOLS[p]))
    B, C, T, H, W = x.shape
    x = x.permute(0, 1, 3, 4, 2).reshape(B, C * H * W, T)                  # b c t h w -> b (c h w) t
    p = "block3.5"
    return F.leaky_relu(_bn(sd, p, F.conv1d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"])), 0.01)


def prepare_chunks(frames_u8, chunk=512, side=96):
    """uint8 [T, H, W, 3] -> list of [3, chunk, side, side] (VideoData.__getitem__)."""
    video = frames_u8.permute(0, 3, 1, 2) / 255
    n = video.shape[0]
    out = []
    for i in range(n // chunk + 1):
        sub = video[i * chunk:(i + 1) * chunk]
        if sub.shape[0] < chunk:
            sub = F.pad(sub, [0, 0, 0, 0, 0, 0, 0, chunk - sub.shape[0]])
        out.append(F.interpolate(sub, size=(side, side), mode="nearest").permute(1, 0, 2, 3))
    return out


def video_features(sd, frames_u8, chunk=512):
    n = frames_u8.shape[0]
    chunks = prepare_chunks(frames_u8, chunk)
    feats = encoder_forward(sd, torch.stack(chunks)).permute(0, 2, 1)
    rows = [feats[i] for i in range(len(chunks) - 1)] + [feats[len(chunks) - 1, :n % chunk]]
    return torch.cat(rows, 0)
