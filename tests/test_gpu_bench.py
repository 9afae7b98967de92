"""bench.py as the driver runs it: the single-GPU line, the RCCL code path, and the self-launched multi-rank path.

A one-GPU box cannot hold two RCCL ranks (one communicator per device), so RCCL itself is exercised with ONE rank
(`BCBF_BENCH_FORCE_DIST=1`: process group, barriers, the three final collectives all go through ncclComm on the
GPU), and the N = 2 launcher path with two ranks sharing cuda:0 over gloo (`BCBF_BENCH_SINGLE_DEVICE=1`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "4", "--warmup", "2", "--batch", "256", "--ntrain", "128"]


def _run(extra_args, extra_env, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra_args, env=env,
                         capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    out = _run(["--cpu-sample", "8"], {})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["dtype"] == "f32" and out["vs_baseline"] is None
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and rf["unit"] == "GB/s" and "traffic_source" in rf
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-12
    # kernel time <= step time; bytes/step over the step time cannot beat the kernel's own rate
    assert rf["kernel_ms"] <= out["ms_per_step"] * 1.05 and rf["kernel_busy_ms_per_step"] <= out["ms_per_step"] * 1.05
    # achieved = all launches' bytes over the time at least one of them ran
    total = rf["algorithmic_bytes_per_launch"] * rf["launches_per_step"]
    assert abs(total / (rf["kernel_busy_ms_per_step"] * 1e-3) / 1e9 - rf["achieved"]) < 1e-6 * rf["achieved"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0


def test_rccl_path_with_one_rank():
    """ncclCommInit + barrier + all_reduce(MAX/SUM) + all_gather on the GPU box."""
    out = _run(["--cpu-sample", "0"], {"BCBF_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                       "MASTER_PORT": "29733"})
    assert out["comm"]["backend"] == "rccl" and out["comm"]["world_size"] == 1 and out["n_gpus"] == 1
    assert len(out["comm"]["per_rank_ms_per_step"]) == 1


def test_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` (no launcher): two child ranks, value = both shards over the slowest rank's time."""
    out = _run(["--gpus", "2", "--cpu-sample", "0"], {"BCBF_BENCH_SINGLE_DEVICE": "1", "BCBF_BENCH_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["comm"]["world_size"] == 2 and out["comm"]["backend"] == "gloo"
    ms = out["comm"]["per_rank_ms_per_step"]
    assert len(ms) == 2 and abs(max(ms) - out["ms_per_step"]) < 1e-6
    assert abs(out["value"] - 2 * 256 * 4 / (out["ms_per_step"] * 4e-3)) / out["value"] < 1e-9


def _run_script(script, args, extra_env, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, env=env, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


TWO_RANKS_ONE_DEVICE = {"BCBF_BENCH_SINGLE_DEVICE": "1", "BCBF_BENCH_BACKEND": "gloo"}


def test_c4_rollouts_two_ranks_equal_one_rank_statistics():
    """BASELINE configs[3] harness: `--gpus 2` starts two ranks itself; each runs its contiguous shard of the trajectories,
    one final reduction: counts add up to the one-rank job's, no collision / solver failure on either, the reported time is
    the slowest rank's; `bench.py --config c4 --gpus 2` is the same harness."""
    args = ["--trajectories", "128", "--steps", "25"]
    one = _run_script("examples_mc_rollouts.py", args, {})
    two = _run_script("examples_mc_rollouts.py", ["--gpus", "2"] + args, TWO_RANKS_ONE_DEVICE)
    assert two["n_gpus"] == 2 and two["comm"]["world_size"] == 2 and two["comm"]["backend"] == "gloo"
    assert len(two["comm"]["per_rank_seconds"]) == 2 and abs(max(two["comm"]["per_rank_seconds"]) - two["seconds"]) < 1e-9
    assert two["count"] == one["count"] == 128 and two["solver_failures"] == one["solver_failures"] == 0
    assert two["collisions"] == one["collisions"] == 0
    assert two["min_h"] > 0 and abs(two["trajectory_steps_per_s"] - 128 * 25 / two["seconds"]) < 1e-6 * two["trajectory_steps_per_s"]
    via_bench = _run_script("bench.py", ["--config", "c4", "--gpus", "2"] + args, TWO_RANKS_ONE_DEVICE)
    assert via_bench["n_gpus"] == 2 and via_bench["count"] == 128
    assert via_bench["steps"] == 25 and via_bench["trajectories"] == 128      # flags bench.py also defines reach the harness


def test_c5_online_growth_two_ranks():
    """BASELINE configs[4] harness with two ranks (weak scaling: `--batch` instances per rank): per-segment times are the
    slowest rank's, failures add up, the end-to-end deviation is the worst rank's."""
    args = ["--batch", "8", "--n0", "40", "--n1", "100"]
    two = _run_script(os.path.join("tools", "bench_online.py"), ["--gpus", "2"] + args, TWO_RANKS_ONE_DEVICE)
    assert two["n_gpus"] == 2 and two["comm"]["world_size"] == 2 and two["scaling"] == "weak" and two["batch_per_gpu"] == 8
    assert two["append_failures"] == 0 and two["refit_failures"] == 0
    assert two["final_vs_refit"]["Mk"] < 1e-8 and two["final_vs_refit"]["Bk"] < 1e-8
    assert [s["N_from"] for s in two["segments"]] == [40] and two["segments"][0]["append_ms"] > 0
    assert abs(two["instance_appends_per_s"] - 60 * 8 * 2 / two["seconds"]) < 1e-6 * two["instance_appends_per_s"]
    one = _run_script("bench.py", ["--config", "c5"] + args, {})
    assert one["n_gpus"] == 1 and one["comm"]["world_size"] == 1 and one["final_vs_refit"]["Mk"] < 1e-8
    assert one["batch_per_gpu"] == 8 and [s["N_from"] for s in one["segments"]] == [40]     # --batch 8 was not swallowed
    for s in two["segments"] + one["segments"]:
        assert abs(s["step_ms"] - (s["append_ms"] + s["control_step_ms"])) < 1e-9


def test_one_gpu_run_through_the_unhooked_launcher_parent(tmp_path):
    """The path `bench.py --gpus 8` takes on an 8-GPU node, on the one GPU there is: NO single-device hook, so the parent
    counts the devices itself (sysfs / throw-away child, never HIP in the parent), holds no descriptor of the GPU driver
    when it spawns torch.distributed.run, and the child rank runs the bench over RCCL (world size 1)."""
    rep = tmp_path / "launch.json"
    out = _run(["--gpus", "1", "--cpu-sample", "0"], {"BCBF_BENCH_FORCE_LAUNCH": "1", "BCBF_BENCH_FORCE_DIST": "1",
                                                      "BCBF_LAUNCH_REPORT": str(rep)})
    r = json.loads(rep.read_text())
    assert r["how"] in ("sysfs", "child") and r["counted"] >= 1 and r["gpu_descriptors"] == [] and r["gpus"] == 1
    assert out["n_gpus"] == 1 and out["comm"]["backend"] == "rccl" and out["comm"]["world_size"] == 1
    assert out["value"] > 0


@pytest.mark.parametrize("config", ["c4", "c5"])
def test_c4_c5_through_the_unhooked_launcher_parent_one_rccl_rank(tmp_path, config):
    """`bench.py --config c4|c5 --gpus 8` on an 8-GPU node takes this path; here with the one GPU there is: the parent counts
    devices without HIP, holds no GPU descriptor when it spawns torch.distributed.run, and the ONE child rank runs the
    harness over RCCL (communicator, barriers, the final reductions)."""
    rep = tmp_path / "launch.json"
    args = (["--trajectories", "64", "--steps", "10"] if config == "c4" else ["--batch", "8", "--n0", "40", "--n1", "72"])
    out = _run_script("bench.py", ["--config", config, "--gpus", "1"] + args,
                      {"BCBF_BENCH_FORCE_LAUNCH": "1", "BCBF_BENCH_FORCE_DIST": "1", "BCBF_LAUNCH_REPORT": str(rep)})
    r = json.loads(rep.read_text())
    assert r["how"] in ("sysfs", "child") and r["counted"] >= 1 and r["gpu_descriptors"] == [] and r["gpus"] == 1
    assert out["n_gpus"] == 1 and out["comm"]["backend"] == "rccl" and out["comm"]["world_size"] == 1
    if config == "c4":
        assert out["count"] == 64 and out["solver_failures"] == 0 and out["roofline"]["achieved"] > 0
    else:
        assert out["append_failures"] == 0 and out["final_vs_refit"]["Mk"] < 1e-8
        assert out["segments"][0]["roofline"]["bound"] == "hbm" and 0 < out["segments"][0]["roofline"]["frac"] < 1


def test_learning_loop_two_ranks_and_through_bench_py():
    """The learning closed loop harness (`bench.py --config learn` = tools/bench_learning_loop.py) with two ranks (weak scaling:
    `--batch` instances per rank, no collective inside the loop, times = the slowest rank's) and with one rank through bench.py;
    both schedules; the line carries the shares and a roofline entry per kernel."""
    args = ["--data", "synthetic", "--batch", "16", "--max-train", "96", "--steps", "16", "--warmup", "8", "--refit-every", "8", "--dtype", "f64"]
    two = _run_script(os.path.join("tools", "bench_learning_loop.py"), ["--gpus", "2"] + args, TWO_RANKS_ONE_DEVICE)
    assert two["n_gpus"] == 2 and two["comm"]["world_size"] == 2 and two["scaling"] == "weak" and two["batch"] == 16
    assert two["append_or_refit_failures"] == 0 and two["shares"]["refits_in_timed_region"] == 2
    assert abs(two["value"] - 2 * 16 * 16 / two["seconds"]) < 1e-6 * two["value"]
    assert two["final_vs_fp64_refit_on_device"]["Mk"] < 1e-8 and two["final_vs_fp64_refit_on_device"]["Bk"] < 1e-8
    one = _run_script("bench.py", ["--config", "learn", "--schedule", "reference", "--parts", "2"] + args, {})
    assert one["n_gpus"] == 1 and one["schedule"] == "reference" and one["parts"] == 2 and one["max_train"] == 96
    assert one["roofline"]["pass"]["bound"] == "hbm" and one["roofline"]["refit"]["bound"] == "mfma"
    assert one["final_vs_fp64_refit_on_device"]["Mk"] < 1e-8
    assert two["schedule"] == "online_tail"                       # (the tool's default)
    inp = _run_script(os.path.join("tools", "bench_learning_loop.py"), ["--schedule", "online", "--parts", "2"] + args, {})
    assert inp["schedule"] == "online" and inp["parts"] == 2 and inp["append_or_refit_failures"] == 0
    assert inp["final_vs_fp64_refit_on_device"]["Mk"] < 1e-8 and inp["final_vs_fp64_refit_on_device"]["Bk"] < 1e-8
    # the loop that learns from ITSELF (the tool's default data): two ranks, and one rank through bench.py with a hyper-parameter fit
    largs = ["--batch", "16", "--max-train", "96", "--steps", "16", "--refit-every", "8", "--dtype", "f64", "--parts", "2"]
    two = _run_script(os.path.join("tools", "bench_learning_loop.py"), ["--gpus", "2", "--schedule", "reference"] + largs, TWO_RANKS_ONE_DEVICE)
    assert two["n_gpus"] == 2 and two["data"].endswith("(x_t, u_t, x_t+1)") and two["schedule"] == "reference" and two["stagger"]
    assert two["append_or_refit_failures"] == 0 and two["warmup"] >= 96 and two["final_vs_fp64_refit_on_device"]["Mk"] < 1e-8
    one = _run_script("bench.py", ["--config", "learn", "--schedule", "online_tail"] + largs, {})
    assert one["schedule"] == "online_tail" and one["append_or_refit_failures"] == 0 and one["final_vs_fp64_refit_on_device"]["Mk"] < 1e-7
    fit = _run_script("bench.py", ["--config", "learn", "--schedule", "reference", "--fit-iters", "3"] + largs, {})
    assert fit["fit_iters"] == 3 and "3 Adam iterations" in fit["metric"] and fit["append_or_refit_failures"] == 0
    assert fit["final_vs_fp64_refit_on_device"]["Mk"] < 1e-7
