"""`python bench.py --gpus N` with no rendezvous in the environment starts its N ranks itself (round-3 verdict: the
driver's plain command silently ran one process).  On this GPU-less host the launch, rendezvous (127.0.0.1), sharding
and reporting protocol are rehearsed over gloo (AVDF_BENCH_REHEARSE=1: no device work); the RCCL form of the same
launch refuses to start when fewer than N devices are visible instead of running a smaller job under the N label."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True,
                          timeout=300)


def test_gpus_2_self_launches_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--batch", "8"], AVDF_DIST_BACKEND="gloo", AVDF_BENCH_REHEARSE="1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["global_batch"] == 16 and line["rank_sum"] == 1
    assert line["config"]["parallelism"] == "dp2" and abs(line["max_over_ranks_s"] - 0.002) < 1e-12


def test_gpus_n_refuses_without_n_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a host with fewer than 2 GPUs")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and "needs 2 visible GPUs" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", AVDF_BENCH_REHEARSE="1", AVDF_DIST_BACKEND="gloo")
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr
