"""N>1 path of bench.py on CPU: two gloo ranks shard a batch of frames with no data-path
collective; only the timing reduce is collective.  (The GPU work itself is covered by -m gpu.)"""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
shard = bench.shard_frames(total_frames=10, rank=rank, world=world)
t = torch.tensor([0.5 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
out = bench.aggregate(frames_per_rank=[len(bench.shard_frames(10, r, world)) for r in range(world)], steps=3, seconds=float(t.item()))
print(json.dumps({"rank": rank, "shard": list(shard), "max_t": float(t.item()), "value": out}))
dist.destroy_process_group()
'''


def test_two_rank_sharding(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % {"root": ROOT})
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=120)
        assert p.returncode == 0
        outs.append(eval(o.strip().splitlines()[-1].replace("true", "True")))
    shards = sorted(tuple(o["shard"]) for o in outs)
    assert shards == [(0, 1, 2, 3, 4), (5, 6, 7, 8, 9)]          # contiguous blocks, disjoint, complete
    assert all(abs(o["max_t"] - 1.5) < 1e-9 for o in outs)          # MAX over ranks
    assert all(abs(o["value"] - 10 * 3 / 1.5) < 1e-9 for o in outs)  # all ranks' frames / max time
