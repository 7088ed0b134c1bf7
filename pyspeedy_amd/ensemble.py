"""Ensemble sharding across the GPUs of a node.

Members are fully independent (reference: `parallel_step`, registry/templates/speedy_driver.f90.j2:58-79, an OpenMP loop
over members that exchange nothing), so the multi-GPU path is: one process per GPU, block partition of the members,
no data-path collective.  `torch.distributed` (RCCL on the GPU box, gloo in the CPU tests) is used only for the barrier
and for combining timings / counts.
"""
import os


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


def init_process_group(backend, device=None):
    """Initialise torch.distributed when WORLD_SIZE > 1; returns the module or None."""
    world, _, _ = dist_env()
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
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
