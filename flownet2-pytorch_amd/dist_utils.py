"""Multi-GPU plumbing for the custom-layer hot path: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The three layers have no parameters and every output element depends on one batch item only
(reference correlation_cuda_kernel.cu:89,162,255; resample2d_kernel.cu:35; channelnorm_kernel.cu:41),
so the path shards over the batch with no data-path collective.  What the reference does with
single-process nn.DataParallel (main.py:187-201: scatter the batch, broadcast the parameters every
step) reduces here to: each rank takes a contiguous slice of the batch, model weights (if a model
sits around the layers) are broadcast once, and timing / throughput are combined with one MAX
all-reduce.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (set by
    torch.distributed.run).  Returns (rank, world, local_rank); a single process needs no group --
    unless `force` (or FN2_DIST_FORCE_GROUP=1) asks for a one-rank group, which sends every collective
    below through the backend all the same (the 1-GPU boxes' way of executing the RCCL call sites)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force = force or os.environ.get("FN2_DIST_FORCE_GROUP") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        init_file = os.environ.get("FN2_INIT_FILE")
        if init_file:
            # rendezvous through a file instead of a TCP port somebody else may take between choosing and binding it
            # (bench.py's self-launch); RCCL's own bootstrap picks its sockets itself
            kwargs.update(init_method="file://" + init_file, rank=rank, world_size=world)
        dist.init_process_group(backend, **kwargs)
    return rank, world, local_rank


def shard_bounds(total, world, rank):
    """Contiguous, balanced [lo, hi) slice of `total` batch items for `rank` (first total % world
    ranks get one extra item)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(tensors, world, rank):
    """Slice every tensor of a batch along dim 0 for this rank (views, no copy)."""
    out = []
    for t in tensors:
        lo, hi = shard_bounds(t.shape[0], world, rank)
        out.append(t[lo:hi])
    return out


def broadcast_state(module_or_tensors, src=0):
    """One-time broadcast of parameters/buffers from `src` (the reference's DataParallel re-broadcasts
    them every forward).  Accepts an nn.Module or an iterable of tensors; a single flat buffer per
    dtype keeps it to a few large collectives (xGMI is per-link bound: few, big messages)."""
    if not dist.is_initialized():
        return
    if isinstance(module_or_tensors, torch.nn.Module):
        tensors = [p.data for p in module_or_tensors.parameters()] + [b.data for b in module_or_tensors.buffers()]
    else:
        tensors = list(module_or_tensors)
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for group in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (step time is the slowest rank's)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_throughput(items_this_rank, elapsed_this_rank, device=None):
    """Units processed by all ranks / slowest rank's time."""
    return sum_over_ranks(items_this_rank, device) / max_over_ranks(elapsed_this_rank, device)
