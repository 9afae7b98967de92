"""Multi-GPU: instances shard contiguously over ranks (one process per GPU), no collective inside
the control loop, one all-reduce of a small statistics vector at the end.
Backend-agnostic: `nccl` (= RCCL over xGMI) on the GPUs, `gloo` in the CPU tests."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous split of `total` instances; the first `total % world` ranks get one more."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard(tensor, rank, world, dim=0):
    a, b = shard_range(tensor.shape[dim], rank, world)
    return tensor.narrow(dim, a, b - a)


def reduce_rollout_stats(collisions, min_h, cost_sum, solver_failures, count):
    """Final reduction of per-shard Monte-Carlo statistics: SUM for counts / costs, MIN for the
    smallest barrier value seen.  Inputs are python numbers or 0-d tensors on the rank's device."""
    dev = min_h.device if torch.is_tensor(min_h) else "cpu"
    if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
        dev = "cpu"                                   # gloo (CPU tests, one-GPU boxes): host tensors
    sums = torch.tensor([float(collisions), float(cost_sum), float(solver_failures), float(count)],
                        dtype=torch.float64, device=dev)
    mins = torch.tensor([float(min_h)], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(mins, op=dist.ReduceOp.MIN)
    c, cost, fails, n = sums.tolist()
    return dict(collisions=int(c), mean_cost=cost / max(n, 1.0), solver_failures=int(fails), count=int(n),
                min_h=float(mins[0]))


# ---------------------------------------------------------------------------------------------------------------------
# One launcher for every multi-GPU harness (bench.py = config 3, examples_mc_rollouts.py = config 4,
# tools/bench_online.py = config 5): `python <script> --gpus N` starts N ranks itself, or runs as one rank of an external
# `python -m torch.distributed.run ... <script> --gpus N`.
def launch_ranks(script, argv, gpus):
    """Start `gpus` ranks of `script argv` (one process per GPU) under torch.distributed.run as a CHILD process and return
    its exit code.  The calling parent never touches the GPU (no HIP call before or after; counting devices does not
    initialise the runtime) and never replaces itself (no exec).  Refuses (exit code 2) when fewer GPUs are visible,
    unless BCBF_BENCH_SINGLE_DEVICE=1 (test hook: every rank on cuda:0)."""
    if os.environ.get("BCBF_BENCH_SINGLE_DEVICE") != "1":
        have = torch.cuda.device_count()
        if have < gpus:
            sys.stderr.write("%s: --gpus %d but only %d GPU(s) visible\n" % (os.path.basename(script), gpus, have))
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


class RankContext:
    """This process as one rank of a one-process-per-GPU job: reads RANK / LOCAL_RANK / WORLD_SIZE, binds the device,
    creates the process group (RCCL = backend "nccl"; over xGMI on a multi-GPU node).  Test hooks for one-GPU boxes:
    BCBF_BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0, BCBF_BENCH_BACKEND=gloo replaces RCCL (two RCCL ranks cannot
    share a device), BCBF_BENCH_FORCE_DIST=1 runs the N > 1 code path (communicator, barriers, the final reductions over
    RCCL) with a single rank too."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("BCBF_BENCH_SINGLE_DEVICE") == "1":
            local_rank = 0
        self.backend = os.environ.get("BCBF_BENCH_BACKEND", "nccl")
        self.multi = self.world > 1 or os.environ.get("BCBF_BENCH_FORCE_DIST") == "1"
        torch.cuda.set_device(local_rank if self.multi else 0)
        if self.multi:
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend=self.backend)
        self.device = torch.device("cuda", torch.cuda.current_device())

    @property
    def backend_name(self):
        return (("rccl" if self.backend == "nccl" else self.backend) if self.multi else None)

    def barrier(self):
        if self.multi:
            dist.barrier()

    def reduce_times(self, seconds):
        """(max over ranks, [every rank's own seconds]) -- the job's time is its slowest rank's."""
        t = torch.tensor([float(seconds)], dtype=torch.float64, device=self.device)
        if not self.multi:
            return float(seconds), [float(seconds)]
        if self.backend != "nccl":
            t = t.cpu()
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(per, t)
        return float(tmax[0]), [float(v[0]) for v in per]

    def reduce_sum(self, values):
        """SUM of a short list of numbers over the ranks (the final, only data reduction of a job)."""
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self.device)
        if self.multi:
            if self.backend != "nccl":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.tolist()

    def comm_info(self, per_rank_seconds=None, scale=1.0):
        out = {"backend": self.backend_name, "world_size": dist.get_world_size() if self.multi else 1}
        if per_rank_seconds is not None:
            out["per_rank_seconds"] = [s * scale for s in per_rank_seconds]
        return out

    def close(self):
        if self.multi:
            dist.destroy_process_group()
