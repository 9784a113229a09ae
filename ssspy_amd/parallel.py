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

from . import _device as dv


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


def pipeline_blocks(n_mixtures: int, sub_batch: int, ramp: bool = True):
    """The (lo, hi) mixture ranges ``separate_pipelined`` works through, in order.  The first upload
    and the last download have nothing to hide behind: with ``ramp`` (and more than two sub-batches
    of work) a short leading and a short trailing block, a quarter of ``sub_batch``, keep that
    exposed transfer small."""
    B = int(n_mixtures)
    sub_batch = max(1, min(int(sub_batch), B))
    edge = max(1, sub_batch // 4) if ramp and B > 2 * sub_batch else 0
    blocks, lo = [], 0
    if edge:
        blocks.append((0, edge))
        lo = edge
    stop = B - edge if edge else B
    while lo < stop:
        blocks.append((lo, min(lo + sub_batch, stop)))
        lo = blocks[-1][1]
    if edge:
        blocks.append((stop, B))
    return blocks


def separate_pipelined(make_separator: Callable[[], object], X, sub_batch: int, n_iter: int = 100,
                       out: Optional[np.ndarray] = None, ramp: bool = True,
                       **call_kwargs) -> np.ndarray:
    """Separate a HOST-resident batch ``X`` (n_mixtures, n_channels, n_bins, n_frames) complex128
    in sub-batches of ``sub_batch`` mixtures, overlapping the PCIe transfers with the iterations:
    while the separator iterates on sub-batch k (torch's current stream), a copy stream uploads
    sub-batch k + 1 and a second one downloads the result of k - 1.  Mixtures are independent, so
    the result equals ``make_separator()(X, n_iter)`` element for element -- PROVIDED the initial
    state of a mixture does not depend on where it sits in a call: pass it explicitly
    (``basis=`` / ``activation=`` in ``call_kwargs``, per mixture) or accept that a separator seeded
    with one ``rng`` draws a (sub_batch, ...) block per sub-batch where the single call draws one
    (n_mixtures, ...) block, i.e. different random initial values for all but the first block.

    ``make_separator()`` returns a fresh separator (``GaussILRMA`` ...) per sub-batch.  Transfers are
    asynchronous only from / to page-locked memory: a pinned ``X`` (e.g. the ``.numpy()`` view of
    ``torch.empty(..., pin_memory=True)``) is uploaded in place, anything else is staged through two
    pinned buffers of one sub-batch each (one extra host copy).  The result is written to ``out``
    (allocated pinned when None) and returned as a NumPy array.
    """
    Xt = X if isinstance(X, torch.Tensor) else torch.from_numpy(X)
    if Xt.dtype != torch.complex128 or Xt.dim() != 4:
        raise ValueError("separate_pipelined: X must be (n_mixtures, n_channels, n_bins, n_frames) complex128")
    B = Xt.shape[0]
    sub_batch = max(1, min(int(sub_batch), B))
    dev = torch.device("cuda", torch.cuda.current_device())
    compute = torch.cuda.current_stream()
    up, down = torch.cuda.Stream(), torch.cuda.Stream()
    if out is None:
        out_t = torch.empty(tuple(Xt.shape), dtype=torch.complex128, pin_memory=True)
    else:
        out_t = out if isinstance(out, torch.Tensor) else torch.from_numpy(out)
    direct_in, direct_out = Xt.is_pinned(), out_t.is_pinned()
    shape1 = (sub_batch,) + tuple(Xt.shape[1:])  # (edge blocks are smaller: they fit)
    stage_in = None if direct_in else [torch.empty(shape1, dtype=torch.complex128, pin_memory=True)
                                       for _ in range(2)]
    stage_out = None if direct_out else [torch.empty(shape1, dtype=torch.complex128, pin_memory=True)
                                         for _ in range(2)]
    blocks = pipeline_blocks(B, sub_batch, ramp)

    slot_read = [None, None]  # per input staging slot: the event of the upload that last read it

    def upload(k):
        lo, hi = blocks[k]
        src = Xt[lo:hi]
        if not direct_in:
            if slot_read[k % 2] is not None:
                slot_read[k % 2].synchronize()  # upload k - 2 has left the slot
            stage_in[k % 2][: hi - lo].copy_(src)  # host copy
            src = stage_in[k % 2][: hi - lo]
        with torch.cuda.stream(up):
            xd = src.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(up)
        slot_read[k % 2] = ev
        return xd, ev

    pending = []  # (block index, device result, download-finished event, staging slot or None)

    def retire(entry):
        k, yd, ev, slot = entry
        ev.synchronize()
        if slot is not None:
            lo, hi = blocks[k]
            out_t[lo:hi].copy_(stage_out[slot][: hi - lo])

    nxt = upload(0)
    for k, (lo, hi) in enumerate(blocks):
        xd, ev_up = nxt
        if k + 1 < len(blocks):
            nxt = upload(k + 1)
        compute.wait_event(ev_up)
        sep = make_separator()
        # the separator's initial state (random basis / activation: host work) is uploaded without
        # waiting for the previous sub-batch, so the host prepares k + 1 while the device iterates on k
        with dv.staged_uploads():
            yd = sep.call_on_device(xd, n_iter=n_iter, **call_kwargs)
        done = torch.cuda.Event()
        done.record(compute)
        # the staging slot of download k was last used by download k - 2: retire it first
        while len(pending) >= 2:
            retire(pending.pop(0))
        with torch.cuda.stream(down):
            down.wait_event(done)
            slot = None if direct_out else k % 2
            dst = out_t[lo:hi] if direct_out else stage_out[slot][: hi - lo]
            dst.copy_(yd, non_blocking=True)
            ev_dn = torch.cuda.Event()
            ev_dn.record(down)
        yd.record_stream(down)
        xd.record_stream(compute)
        pending.append((k, yd, ev_dn, slot))
    for entry in pending:
        retire(entry)
    return out_t.numpy()
