"""World-size-2 gloo test (CPU) of the data-parallel protocol used by engine.TrainStep: clips are
sharded over ranks, the positive count is SUM-reduced before the losses are normalised, gradients
are SUM-reduced -- and the result must equal the single-process full-batch step of the reference
semantics.  The network arithmetic is stood in for by the CPU oracle (the HIP kernels need a GPU);
what is under test is the sharding + the two collectives + the normaliser bookkeeping."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SMALL = dict(input_dim=96, max_seq_len=192, embd_dim=32, n_head=4, fpn_dim=32, head_dim=32)
GKEYS = ["backbone.embd.0.conv.weight", "backbone.stem.0.attn.query.weight", "cls_head.cls_head.conv.bias",
         "reg_head.offset_head.conv.weight", "interpolator.conv1.weight", "backbone.lh_branch.1.ln2.weight"]


def make_batch(cfg):
    rs = np.random.RandomState(5)
    lens = [192, 150, 180, 192]
    x = np.zeros((4, cfg["input_dim"], cfg["max_seq_len"]), np.float32)
    for b, n in enumerate(lens):
        x[b, :, :n] = rs.standard_normal((cfg["input_dim"], n))
    mask = (np.arange(cfg["max_seq_len"])[None] < np.asarray(lens)[:, None]).astype(np.float32)[:, None]
    segs = [torch.tensor([[10.0, 30.0]]), None, torch.tensor([[5.0, 9.0], [100.0, 140.0]]), torch.tensor([[60.0, 64.0]])]
    labels = [None if s is None else torch.zeros(len(s), dtype=torch.int64) for s in segs]
    return torch.from_numpy(x), torch.from_numpy(mask), segs, labels


def run_step(R, cfg, sd, x, mask, segs, labels, num_pos_reduce=None):
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = R.model_forward(sdg, cfg, x, mask, training=True)
    ls, norm, _ = R.losses(cfg, out, segs, labels, cfg["init_loss_norm"], num_pos_reduce=num_pos_reduce)
    ls["final_loss"].backward()
    grads = {k: (sdg[k].grad.clone() if sdg[k].grad is not None else torch.zeros_like(sdg[k])) for k in GKEYS}
    return {k: float(v) for k, v in ls.items()}, norm, grads


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import ummaformer_ref as R
    from audio_visual_deepfake_detection_amd_dp import shard_range, allreduce_sum_
    cfg = R.make_cfg(**SMALL)
    sd = R.synth_state_dict(R.param_shapes(cfg), 3, True)
    x, mask, segs, labels = make_batch(cfg)
    lo, hi = shard_range(4, rank, world)

    def reduce_np(n):
        t = torch.tensor([float(n)])
        allreduce_sum_(t)
        return int(t.item())

    ls, norm, grads = run_step(R, cfg, sd, x[lo:hi], mask[lo:hi], segs[lo:hi], labels[lo:hi], reduce_np)
    for k in GKEYS:
        allreduce_sum_(grads[k])
    tot = torch.tensor([ls["cls_loss"], ls["reg_loss"], ls["reco_cls_loss"]])
    allreduce_sum_(tot)
    if rank == 0:
        full_ls, full_norm, full_grads = run_step(R, cfg, sd, x, mask, segs, labels)
        ret["ok_norm"] = abs(norm - full_norm) < 1e-6
        ret["loss_err"] = max(abs(float(tot[0]) - full_ls["cls_loss"]), abs(float(tot[1]) - full_ls["reg_loss"]),
                              abs(float(tot[2]) - full_ls["reco_cls_loss"]))
        ret["grad_err"] = max(float((grads[k] - full_grads[k]).abs().max() / (full_grads[k].abs().max() + 1e-12))
                              for k in GKEYS)
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    sys.modules.setdefault("audio_visual_deepfake_detection_amd_dp", _load_dp())
    from audio_visual_deepfake_detection_amd_dp import shard_range
    for n in (0, 1, 7, 64, 10000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _load_dp():
    """dp.py has no HIP dependency; load it as a file so the test does not need libavdf_hip.so's device."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "audio_visual_deepfake_detection_amd_dp", os.path.join(ROOT, "audio_visual_deepfake_detection_amd", "dp.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _entry(rank, world, port, ret):
    sys.modules["audio_visual_deepfake_detection_amd_dp"] = _load_dp()
    worker(rank, world, port, ret)


def test_two_rank_step_equals_full_batch_step():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_entry, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok_norm"]
    assert ret["loss_err"] < 1e-5, ret["loss_err"]
    assert ret["grad_err"] < 1e-4, ret["grad_err"]
