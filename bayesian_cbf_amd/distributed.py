"""Multi-GPU: instances shard contiguously over ranks (one process per GPU), no collective inside
the control loop, one all-reduce of a small statistics vector at the end.
Backend-agnostic: `nccl` (= RCCL over xGMI) on the GPUs, `gloo` in the CPU tests."""
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
    sums = torch.tensor([float(collisions), float(cost_sum), float(solver_failures), float(count)],
                        dtype=torch.float64, device=dev)
    mins = torch.tensor([float(min_h)], dtype=torch.float64, device=dev)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(mins, op=dist.ReduceOp.MIN)
    c, cost, fails, n = sums.tolist()
    return dict(collisions=int(c), mean_cost=cost / max(n, 1.0), solver_failures=int(fails), count=int(n),
                min_h=float(mins[0]))
