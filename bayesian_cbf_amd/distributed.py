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
KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def _filter_indices(value, have):
    """Device indices one *_VISIBLE_DEVICES value keeps out of `have` devices, as the runtimes read it: a comma list of
    indices (or UUIDs, which count as present), cut at the first negative or unparsable entry, out-of-range and repeated
    indices dropped."""
    keep = []
    for e in value.split(","):
        e = e.strip()
        if e == "":
            break
        if e.upper().startswith("GPU-"):          # a UUID: cannot be resolved without the runtime; assume it names a device
            keep.append(("uuid", e))
            continue
        try:
            i = int(e)
        except ValueError:
            break
        if i < 0:
            break
        if i < have and i not in keep:
            keep.append(i)
    return keep


def _visible_count(have):
    """Devices left of `have` after the runtime's filters, or None when no filter is set.  ROCR_VISIBLE_DEVICES narrows the
    devices the HSA runtime enumerates; HIP_VISIBLE_DEVICES (or the CUDA spelling torch honours; HIP wins when both are set)
    then indexes INTO that narrowed list -- both apply when both are set."""
    rocr = os.environ.get("ROCR_VISIBLE_DEVICES")
    hip = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    if rocr is None and hip is None:
        return None
    n = have
    if rocr is not None:
        n = len(_filter_indices(rocr, n))
    if hip is not None:
        n = len(_filter_indices(hip, n))
    return n


def visible_gpu_count(nodes_dir=None, dev_dir="/dev"):
    """(count, how): GPUs this user can open, WITHOUT a HIP / HSA call in this process.

    1. the kernel driver's topology in sysfs: a node of /sys/class/kfd/kfd/topology/nodes/*/properties with
       `simd_count` > 0 is a GPU (CPU nodes carry 0); it counts when its render node /dev/dri/renderD<drm_render_minor>
       and /dev/kfd are accessible to this user; narrowed by the *_VISIBLE_DEVICES filter;
    2. where sysfs is not there (containers that only pass /dev/kfd through): a short-lived CHILD python that prints
       `torch.cuda.device_count()` and exits -- whatever the runtime does in there dies with it.
    Never `torch.cuda.device_count()` in this process: without amdsmi it falls back to hipGetDeviceCount, which brings
    the runtime up in a parent that is about to start the ranks."""
    n = None
    try:
        n = 0
        nodes_dir = nodes_dir or KFD_NODES
        for node in sorted(os.listdir(nodes_dir)):
            with open(os.path.join(nodes_dir, node, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            # a container / cgroup may expose fewer devices than the host's topology lists: count a GPU only when its
            # render node is there and this user may open it (an access() check, not an open())
            minor = int(props.get("drm_render_minor", "-1"))
            if minor >= 0 and not os.access(os.path.join(dev_dir, "dri", "renderD%d" % minor), os.R_OK | os.W_OK):
                continue
            n += 1
        if not os.access(os.path.join(dev_dir, "kfd"), os.R_OK | os.W_OK):
            n = 0
        how = "sysfs"
    except (OSError, ValueError):
        n = None
    if n is None:
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                 capture_output=True, text=True, timeout=600)
            n, how = int(out.stdout.strip().splitlines()[-1]), "child"
        except Exception:
            return 0, "unknown"
        return n, how                      # (the child applied the *_VISIBLE_DEVICES filter itself)
    vis = _visible_count(n)
    if vis is not None:
        n = min(n, vis)
    return n, how


def open_gpu_descriptors():
    """Targets of this process's open descriptors that belong to the GPU driver (/dev/kfd, /dev/dri/renderD*): empty as
    long as nothing has initialised the runtime here."""
    out = []
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                tgt = os.readlink("/proc/self/fd/" + fd)
            except OSError:
                continue
            if tgt == "/dev/kfd" or tgt.startswith("/dev/dri/renderD"):
                out.append(tgt)
    except OSError:
        pass
    return out


def launch_ranks(script, argv, gpus):
    """Start `gpus` ranks of `script argv` (one process per GPU) under torch.distributed.run as a CHILD process and return
    its exit code.  The calling parent never touches the GPU -- devices are counted from sysfs or by a throw-away child
    (`visible_gpu_count`), there is no HIP call before or after -- and never replaces itself (no exec).  Refuses (exit
    code 2) when fewer GPUs are visible, unless BCBF_BENCH_SINGLE_DEVICE=1 (test hook: every rank on cuda:0), and (exit
    code 3) when this process already holds the GPU driver open: the ranks must be children of a clean parent.
    BCBF_LAUNCH_REPORT=<path>: what the parent saw at the moment it spawned, as JSON (tests)."""
    have, how = None, "not counted (BCBF_BENCH_SINGLE_DEVICE=1)"
    if os.environ.get("BCBF_BENCH_SINGLE_DEVICE") != "1":
        have, how = visible_gpu_count()
        if have < gpus:
            sys.stderr.write("%s: --gpus %d but only %d GPU(s) visible (%s)\n" % (os.path.basename(script), gpus, have, how))
            return 2
    held = open_gpu_descriptors()
    if held:
        sys.stderr.write("%s: the launcher process has the GPU driver open (%s); start the ranks from a process that has "
                         "not initialised HIP\n" % (os.path.basename(script), ", ".join(sorted(set(held)))))
        return 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(script)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rep = os.environ.get("BCBF_LAUNCH_REPORT")
    if rep:
        import json
        with open(rep, "w") as fh:
            json.dump(dict(pid=os.getpid(), gpus=gpus, counted=have, how=how, gpu_descriptors=held, cmd=cmd), fh)
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
