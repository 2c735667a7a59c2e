"""Tensor-parallel group helpers. One process per GPU; the default process group is the TP group
(backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests). Without an initialised group
every layer behaves as tp_size 1."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import comm as _comm


def tp_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def tp_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _staged(t) -> bool:
    """True when the collective has to go through host memory: a GPU tensor on a "gloo" group,
    the single-GPU rehearsal of the tensor-parallel path (Config.shared_device). Never taken
    with the "nccl" backend."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce(t):
    """In-place sum over the TP group (reference: layers/linear.py:388-389). Decode-sized bf16
    messages (up to 4 MiB) take the direct path over the peers' IPC-mapped buffers when a
    communicator is installed (one kernel, hipGraph-capturable); everything else is RCCL."""
    direct = _comm.active()
    if direct is not None and direct.usable(t):
        return direct.all_reduce(t)
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def gather_to_rank0(t):
    """Rank 0 receives every rank's tensor (list in rank order), the others None
    (reference: layers/embed_head.py:180-192)."""
    rank0 = tp_rank() == 0
    if _staged(t):
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(tp_size())] if rank0 else None
        dist.gather(h, parts, 0)
        return [p.to(t.device) for p in parts] if rank0 else None
    parts = [torch.empty_like(t) for _ in range(tp_size())] if rank0 else None
    dist.gather(t, parts, 0)
    return parts


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0, f"{numerator} is not divisible by {denominator}"
    return numerator // denominator
