"""
dist.py — data-parallel plumbing for the SVI step: one process per GPU, the global
minibatch sharded by contiguous slices, ONE all-reduce(SUM) per step over the flat
gradient buffer with the ELBO scalars in its last 4 slots (RCCL over xGMI when the
backend is "nccl"; gloo in the CPU tests).

The reference has no distributed code; sharding is exact because its
loss is a SUM over the plate (models/ivae.py:177,215): the global gradient is the
sum of the shard gradients, so no averaging is applied.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as td


def world(group=None) -> Tuple[int, int]:
    """(rank, world_size) of the process group, (0, 1) when torch.distributed is not initialised."""
    if td.is_available() and td.is_initialized():
        return td.get_rank(group), td.get_world_size(group)
    return 0, 1


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of a global batch of n samples owned by `rank`.
    The first n % world_size ranks get one extra sample; ranks may get an empty slice."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def allreduce_sum_(flat: torch.Tensor, group=None) -> None:
    """In-place SUM all-reduce of the flat gradient(+scalars) buffer; no-op for world size 1."""
    if td.is_available() and td.is_initialized() and td.get_world_size(group) > 1:
        td.all_reduce(flat, op=td.ReduceOp.SUM, group=group)


def broadcast_(flat: torch.Tensor, src: int = 0, group=None) -> None:
    """Makes every replica start from rank `src`'s parameters."""
    if td.is_available() and td.is_initialized() and td.get_world_size(group) > 1:
        td.broadcast(flat, src=src, group=group)


def sync_replicas(engine, group=None, src: int = 0) -> None:
    """Start of a data-parallel run: every replica takes rank `src`'s parameters — the flat buffer (which also holds
    batch-norm running statistics) AND the parameters of user-defined modules that live outside it
    (set_encoder / set_decoder / set_classifier: engine._enc_params).

    Batch normalisation: per-shard batch statistics are NOT the global batch's, so a sharded step would differ from the
    single-process step the parity bar is defined on, and the running estimates would drift apart between replicas.
    Rejected rather than silently different (the reference has no distributed mode to mirror)."""
    if not (td.is_available() and td.is_initialized() and td.get_world_size(group) > 1):
        return
    if getattr(engine, "_bn_enc", None) or getattr(engine, "_bn_dec", None):
        raise NotImplementedError("batchnorm=True models are not trained data-parallel: per-shard batch statistics "
                                  "differ from the global batch's (train them on one GPU, or build with batchnorm=False)")
    broadcast_(engine.flat, src, group)
    for q in getattr(engine, "_enc_params", None) or []:
        td.broadcast(q.data, src=src, group=group)
