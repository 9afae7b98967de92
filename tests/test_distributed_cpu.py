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


def _fake_topology(tmp_path, simd_counts, minors, present_minors, kfd=True):
    nodes = tmp_path / "nodes"
    dev = tmp_path / "dev"
    (dev / "dri").mkdir(parents=True)
    if kfd:
        (dev / "kfd").write_text("")
    for m in present_minors:
        (dev / "dri" / ("renderD%d" % m)).write_text("")
    for i, (sc, mn) in enumerate(zip(simd_counts, minors)):
        d = nodes / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\ndrm_render_minor %d\n" % (0 if sc else 64, sc, mn))
    return str(nodes), str(dev)


def test_visible_gpu_count_reads_the_driver_topology_not_hip(tmp_path, monkeypatch):
    """The launcher parent counts GPUs from the kfd topology in sysfs: CPU nodes (simd_count 0) do not count, a GPU whose
    render node this user cannot open does not count, the *_VISIBLE_DEVICES filter narrows, and torch.cuda is never asked."""
    from bayesian_cbf_amd import distributed as D
    monkeypatch.setattr(torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("HIP asked in the parent")))
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    nodes, dev = _fake_topology(tmp_path / "a", [0, 0, 1024, 1024, 1024], [-1, -1, 128, 129, 130], [128, 129, 130])
    assert D.visible_gpu_count(nodes, dev) == (3, "sysfs")
    nodes, dev = _fake_topology(tmp_path / "b", [0, 1024, 1024, 1024], [-1, 128, 129, 130], [129])      # one-GPU container
    assert D.visible_gpu_count(nodes, dev) == (1, "sysfs")
    nodes, dev = _fake_topology(tmp_path / "c", [0, 1024], [-1, 128], [128], kfd=False)
    assert D.visible_gpu_count(nodes, dev) == (0, "sysfs")
    nodes, dev = _fake_topology(tmp_path / "d", [1024] * 8, list(range(128, 136)), list(range(128, 136)))
    assert D.visible_gpu_count(nodes, dev) == (8, "sysfs")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3")
    assert D.visible_gpu_count(nodes, dev) == (2, "sysfs")
    # both filters apply: ROCr narrows the enumeration to 4 devices, HIP then indexes into THAT list (index 5 is out of range)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "0,1,2,3")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3,5")
    assert D.visible_gpu_count(nodes, dev) == (2, "sysfs")
    # a negative entry hides everything after it; an unparsable one likewise; repeats count once
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,-1,2,3")
    assert D.visible_gpu_count(nodes, dev) == (1, "sysfs")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "-1")
    assert D.visible_gpu_count(nodes, dev) == (0, "sysfs")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,2,x,4")
    assert D.visible_gpu_count(nodes, dev) == (1, "sysfs")
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "0,1,2,9")
    assert D.visible_gpu_count(nodes, dev) == (3, "sysfs")


_RANK_SCRIPT = """import os, sys
print("rank %s of %s" % (os.environ["RANK"], os.environ["WORLD_SIZE"]), flush=True)
"""

_PARENT = """import json, os, sys
sys.path.insert(0, %(root)r)
from bayesian_cbf_amd import distributed as D
D.visible_gpu_count = lambda *a: (2, "test")          # (this box has no GPU; the count is not what is under test)
%(extra)s
sys.exit(D.launch_ranks(%(script)r, [], 2))
"""


def test_launcher_parent_is_clean_when_it_spawns_and_refuses_when_it_is_not(tmp_path):
    """The path an 8-GPU `bench.py --gpus 8` takes: the parent process spawns torch.distributed.run as a child while
    holding NO descriptor of the GPU driver (/dev/kfd, /dev/dri/renderD*) -- recorded at the moment of the spawn -- and both
    ranks run.  A parent that does hold one refuses with exit code 3 instead of starting ranks."""
    import json
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    rep = tmp_path / "report.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BCBF_BENCH_SINGLE_DEVICE")}
    env["BCBF_LAUNCH_REPORT"] = str(rep)
    res = subprocess.run([sys.executable, "-c", _PARENT % dict(root=ROOT, script=str(script), extra="")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "rank 0 of 2" in res.stdout and "rank 1 of 2" in res.stdout
    r = json.loads(rep.read_text())
    assert r["gpu_descriptors"] == [] and r["counted"] == 2 and r["gpus"] == 2
    assert "torch.distributed.run" in r["cmd"] and "--master-addr" in r["cmd"] and "127.0.0.1" in r["cmd"]
    rep.unlink()
    dirty = "D.open_gpu_descriptors = lambda: ['/dev/kfd']"
    res = subprocess.run([sys.executable, "-c", _PARENT % dict(root=ROOT, script=str(script), extra=dirty)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 3 and "GPU driver open" in res.stderr and not rep.exists() and "rank 0" not in res.stdout


def test_open_gpu_descriptors_sees_a_held_device_node(tmp_path, monkeypatch):
    """open_gpu_descriptors() really reads /proc/self/fd: a descriptor whose target is named like a render node shows up."""
    from bayesian_cbf_amd import distributed as D
    assert D.open_gpu_descriptors() == []
    real = os.readlink
    held = open(os.devnull)
    monkeypatch.setattr(os, "readlink", lambda p: "/dev/dri/renderD128" if p.endswith("/%d" % held.fileno()) else real(p))
    assert D.open_gpu_descriptors() == ["/dev/dri/renderD128"]
    held.close()
