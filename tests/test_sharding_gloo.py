"""N>1 path of bench.py on CPU.  `python bench.py --gpus 2 --dry-run` goes through the real entry point: the
parent spawns two rank processes (before importing torch), each initialises gloo, takes its contiguous shard of
the global batch (no data-path collective), the ranks barrier, the time is MAX-reduced and rank 0 prints the
line.  Without --dry-run the same launch on a machine with fewer GPUs than ranks must fail loudly instead of
silently measuring one rank.  (The GPU work itself is covered by -m gpu.)"""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_two_ranks_through_the_real_entry_point():
    r = run(["--gpus", "2", "--dry-run", "--steps", "4", "--batch", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["frames_per_rank"] == [5, 5] and line["scaling"] == "weak"
    # whole-job value: both ranks' frames over the slowest rank's time (rank 1 sleeps 2 ms per step)
    assert 0 < line["value"] <= 2 * 5 * 4 / (4 * 0.002)
    # every rank's own time is in the line (a straggler must show: rank 1 sleeps twice as long as rank 0), the divisor of
    # `value` is the slowest one, and the communicator saw both ranks
    pr = line["per_rank_ms_per_step"]
    assert len(pr) == 2 and pr[1] > pr[0] > 0 and line["world_size_observed"] == 2
    assert max(pr) <= line["ms_per_step"] * 1.001


def test_sharding_helpers():
    sys.path.insert(0, ROOT)
    import bench
    shards = [tuple(bench.shard_frames(10, r, 2)) for r in range(2)]
    assert shards == [(0, 1, 2, 3, 4), (5, 6, 7, 8, 9)]          # contiguous blocks, disjoint, complete
    assert [len(bench.shard_frames(7, r, 4)) for r in range(4)] == [2, 2, 2, 1]
    assert abs(bench.aggregate([5, 5], steps=3, seconds=1.5) - 20.0) < 1e-12


def test_launcher_mismatch_and_missing_devices_fail_loudly():
    # ranks already launched by someone else, but not as many as --gpus says
    r = run(["--gpus", "2", "--dry-run"], env={"RANK": "0", "WORLD_SIZE": "3"})
    assert r.returncode != 0 and "WORLD_SIZE is 3" in (r.stderr + r.stdout)
    # two ranks on a machine without two GPUs (this container has none): every rank refuses
    import torch
    if torch.cuda.device_count() < 2:
        r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--no-cpu-baseline"])
        assert r.returncode != 0
        assert "has no GPU of its own" in r.stderr or "ranks requested" in r.stderr, r.stderr[-1500:]
