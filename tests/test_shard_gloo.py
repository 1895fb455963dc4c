"""The multi-GPU path is region sharding with no data-path collective; ranks meet only for the barrier and
the max-over-ranks clock.  Rehearsed here on CPU with gloo, world_size 2."""
import json
import os
import subprocess
import sys

from uvc_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_shards_covers_every_region_once_and_balances():
    costs = [(1000 + 37 * i % 900, 5000) for i in range(41)]
    for world in (1, 2, 4, 8):
        plan = shard.plan_shards(costs, world)
        flat = sorted(i for p in plan for i in p)
        assert flat == list(range(len(costs)))
        assert all(p == sorted(p) for p in plan)
        loads = [sum(costs[i][0] * 2 + costs[i][1] for i in p) for p in plan]
        assert max(loads) - min(loads) <= max(c[0] * 2 + c[1] for c in costs)


def test_two_rank_protocol_with_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0", "--tile-kb", "10", "--dry-run"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # only rank 0 prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 3
    # max over ranks: rank 1 sleeps 20 ms per step, rank 0 only 10 ms
    assert j["ms_per_step"] >= 19.0
    assert abs(j["value"] - 2 * 10000 * 3 / (j["ms_per_step"] * 3 / 1e3)) / j["value"] < 1e-6
