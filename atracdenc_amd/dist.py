"""Multi-GPU plumbing: streams are independent, so a node scales by sharding streams across one process per
GPU. There is no data-path collective; torch.distributed (gloo, CPU side) is used
only for the start/stop barrier, the MAX of the per-rank elapsed time and - in tests - gathering checksums."""
import os


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_streams(n_streams_total, world, rank):
    """Contiguous, balanced partition of stream indices: returns (first, count) for `rank`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_streams_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def init(backend="gloo"):
    """CPU-side process group (gloo): the data path has no exchange step, so nothing ever runs over RCCL."""
    import torch.distributed as dist
    dist.init_process_group(backend=backend)
    return dist


def max_over_ranks(value, dist=None, device="cpu"):
    """MAX all-reduce of a python float (elapsed seconds)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj, dist=None):
    """All ranks' python objects in rank order (tests / small metadata only)."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
