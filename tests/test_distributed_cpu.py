"""world_size-2 gloo test of the sharding + final-reduction logic (the only collective of the path)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    from bayesian_cbf_amd.distributed import shard, shard_range, reduce_rollout_stats
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    h = torch.randn(total, generator=g)                  # per-trajectory minimum barrier value
    cost = torch.rand(total, generator=g)
    fails = (torch.rand(total, generator=g) < 0.1)
    a, b = shard_range(total, rank, world)
    hs, cs, fs = shard(h, rank, world), shard(cost, rank, world), shard(fails, rank, world)
    assert hs.shape[0] == b - a
    out = reduce_rollout_stats((hs < 0).sum(), hs.min(), cs.sum(), fs.sum(), hs.numel())
    ref = dict(collisions=int((h < 0).sum()), mean_cost=float(cost.mean()), solver_failures=int(fails.sum()),
               count=total, min_h=float(h.min()))
    ok = all(abs(out[k] - ref[k]) < 1e-6 for k in ref)
    q.put((rank, ok, (a, b)))
    dist.destroy_process_group()


def test_shard_and_final_reduction_world2():
    world, total = 2, 1001                      # ragged split: 501 + 500
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29611, total, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == (0, 501) and res[1][2] == (501, 1001)


def test_shard_range_covers_everything():
    from bayesian_cbf_amd.distributed import shard_range
    for total in (0, 1, 7, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))


def test_bench_refuses_more_ranks_than_gpus_without_touching_a_gpu():
    """`python bench.py --gpus 2` on a box without 2 GPUs: the launcher parent exits 2 before any rank starts."""
    import subprocess
    import pytest
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible: `bench.py --gpus 2` would run the full workload")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for script, extra in (("bench.py", []), ("bench.py", ["--config", "c4"]), ("bench.py", ["--config", "c5"]),
                          ("examples_mc_rollouts.py", []), (os.path.join("tools", "bench_online.py"), [])):
        res = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2"] + extra, env=env, capture_output=True,
                             text=True, timeout=300)
        assert res.returncode == 2 and "only" in res.stderr, (script, extra, res.returncode, res.stderr[-500:])
