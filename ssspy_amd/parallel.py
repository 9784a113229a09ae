"""Sharding of a batch of independent mixtures across the GPUs of one node.

The demixing path has no cross-mixture coupling, so multi-GPU execution is one process per GPU
(``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` in CPU tests),
each owning a contiguous block of mixtures, with NO collective inside the iteration loop.  The
only communication is optional: gathering per-mixture results and the max-over-ranks of a timing.
"""

import os
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*.

    Returns (rank, world_size, local_rank).  A single process (WORLD_SIZE unset or 1) needs no group.
    """
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n_items` owned by `rank`; the first n_items % world ranks get
    one extra item (configs[4]: 1024 mixtures over 8 GPUs -> 128 each)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_sharded(process_shard: Callable[[int, int], np.ndarray], n_mixtures: int,
                gather: bool = True):
    """Run ``process_shard(lo, hi)`` on this rank's block of mixtures.

    ``process_shard`` returns an array whose leading axis is ``hi - lo`` (one entry per mixture).
    With ``gather`` every rank receives the concatenation over ranks in mixture order (object
    all-gather; meant for filters / losses -- leave large spectrograms sharded with gather=False).
    """
    rank, world = rank_world()
    lo, hi = shard_bounds(n_mixtures, rank, world)
    # more ranks than mixtures: the surplus ranks own an empty block, launch nothing (the C ABI
    # rejects an empty batch) and still take part in the gather
    local = process_shard(lo, hi) if hi > lo else None
    if not gather or world == 1:
        return local
    parts = [None] * world
    dist.all_gather_object(parts, None if local is None else np.asarray(local))
    parts = [p for p in parts if p is not None and p.shape[0] > 0]
    if not parts:
        return np.empty((0,))
    return np.concatenate(parts, axis=0)


def max_over_ranks(seconds: float, device: Optional[torch.device] = None) -> float:
    """Maximum of a per-rank wall time (the job finishes when the slowest rank does)."""
    rank, world = rank_world()
    if world == 1:
        return float(seconds)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) \
            if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
