"""Ensemble sharding across the GPUs of a node.

Members are fully independent (reference: `parallel_step`, registry/templates/speedy_driver.f90.j2:58-79, an OpenMP loop
over members that exchange nothing), so the multi-GPU path is: one process per GPU, block partition of the members,
no data-path collective.  `torch.distributed` (RCCL on the GPU box, gloo in the CPU tests) is used for the barrier, for
combining timings / counts, for the ONE broadcast of the shared boundary fields at start-up (rank 0
reads the file, ~3.4 MB go out over xGMI) and for ensemble statistics (mean / spread over the members of all ranks).
"""
import os

import numpy as np


def shard_members(total_members, world_size, rank):
    """Block partition: returns (first_member, count) owned by `rank`.

    The first `total % world` ranks own one extra member, so counts differ by at most one."""
    if total_members < 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad sharding request: total=%r world=%r rank=%r" % (total_members, world_size, rank))
    base, extra = divmod(total_members, world_size)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def dist_env():
    """(world_size, rank, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend, device=None, force=False):
    """Initialise torch.distributed when WORLD_SIZE > 1 (or when `force` asks for a one-rank group); returns the module or
    None.  The rendezvous address comes from the launcher (torchrun, bench.py's own launcher: MASTER_ADDR / MASTER_PORT);
    there is no built-in port: two jobs on one host must not meet on a hard-wired one."""
    world, _, _ = dist_env()
    if world <= 1 and not force:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            raise RuntimeError("MASTER_PORT is not set: start the ranks with torch.distributed.run or `bench.py --gpus N`")
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    kwargs = {}
    if device is not None and backend == "nccl":
        kwargs["device_id"] = device
    dist.init_process_group(backend, **kwargs)
    return dist


def max_over_ranks(value, dist, device="cpu"):
    """Slowest rank's value (the job's wall time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist, device="cpu"):
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def simulated_years_per_day(total_members, seconds_per_step, steps_per_year=36 * 365):
    """Whole-job throughput: every step advances `total_members` members by 40 model minutes."""
    return total_members * 86400.0 / (seconds_per_step * steps_per_year)


def broadcast_boundary_conditions(fields, dist, device="cpu", src=0):
    """Boundary fields (mapping name -> array, e.g. the 12 example_bc fields [+ ssta]) from rank `src` to every rank.

    Rank `src` passes the mapping, the others pass None.  One small object broadcast carries names / shapes, then ONE
    flat float64 buffer carries all fields (a single collective instead of one per field: xGMI collectives are
    latency-bound at this size).  Returns the mapping (numpy float64 arrays) on every rank."""
    if dist is None:
        return {k: np.asarray(v, dtype=np.float64) for k, v in fields.items()}
    import torch
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        names = sorted(fields)
        meta[0] = [(k, tuple(np.shape(fields[k]))) for k in names]
    dist.broadcast_object_list(meta, src=src, device=torch.device(device) if str(device) != "cpu" else None)
    total = sum(int(np.prod(shape)) for _, shape in meta[0])
    if rank == src:
        flat = np.concatenate([np.asarray(fields[k], dtype=np.float64).ravel() for k, _ in meta[0]])
        buf = torch.from_numpy(flat).to(device)
    else:
        buf = torch.empty(total, dtype=torch.float64, device=device)
    dist.broadcast(buf, src=src)
    host = buf.cpu().numpy()
    out, off = {}, 0
    for k, shape in meta[0]:
        n = int(np.prod(shape))
        out[k] = host[off:off + n].reshape(shape).copy()
        off += n
    return out


def ensemble_mean_spread(values, dist, ddof=1):
    """Mean and spread (standard deviation over members) of a field across ALL members of all ranks.

    values: torch tensor [local_members, ...] (device tensor on the GPU path).  Two all-reduces of one field each:
    member count + sum, then the sum of squared deviations from the global mean (two-pass, fp64)."""
    import torch
    v = values.to(torch.float64)
    packed = torch.cat([v.sum(dim=0).reshape(-1), torch.tensor([float(v.shape[0])], dtype=torch.float64, device=v.device)])
    if dist is not None:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    n = packed[-1]
    mean = (packed[:-1] / n).reshape(v.shape[1:])
    ssq = ((v - mean) ** 2).sum(dim=0)
    if dist is not None:
        dist.all_reduce(ssq, op=dist.ReduceOp.SUM)
    spread = torch.sqrt(ssq / torch.clamp(n - ddof, min=1.0))
    return mean, spread
