"""Tensor-parallel group helpers. One process per GPU; the default process group is the TP group
(backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests). Without an initialised group
every layer behaves as tp_size 1."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import comm as _comm
from . import fused as _fused


def tp_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def tp_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


# Callers whose process group cannot carry the tensors as they are (a harness that rehearses the
# GPU path over a host-only group) install their own carrier for what the direct path does not
# take: an object with all_reduce(t) and gather_to_rank0(t). None: torch.distributed as is.
_CARRIER = None


def set_carrier(carrier) -> None:
    global _CARRIER
    _CARRIER = carrier


def all_reduce(t):
    """In-place sum over the TP group (reference: layers/linear.py:388-389). Decode-sized bf16
    messages (up to 8 MiB: every decode batch the graphs cover, 512 x 8192) take the direct path over the peers' IPC-mapped buffers when a
    communicator is installed (one kernel, hipGraph-capturable); everything else is RCCL."""
    direct = _comm.active()
    if direct is not None and direct.usable(t):
        return direct.all_reduce(t)
    if _CARRIER is not None:
        return _CARRIER.all_reduce(t)
    dist.all_reduce(t)
    return t


def all_reduce_add_rms_norm(x, residual, norm):
    """norm(all_reduce(x), residual) for a RowParallelLinear output that has not been reduced yet
    (reference: layers/linear.py:388-389 followed by layers/layernorm.py:53-80). One kernel on the
    direct path for messages it serves in one shot; otherwise the two steps one after the other.
    Returns (normed x, new residual)."""
    if tp_size() > 1:
        direct = _comm.active()
        if direct is not None and _fused.use_fused(x) and direct.usable_fused(x, residual, norm.weight):
            return direct.all_reduce_add_rms_norm(x, residual, norm.weight, norm.eps)
        all_reduce(x)
    return norm(x, residual)


def gather_to_rank0(t):
    """Rank 0 receives every rank's tensor (list in rank order), the others None
    (reference: layers/embed_head.py:180-192)."""
    if _CARRIER is not None:
        return _CARRIER.gather_to_rank0(t)
    # An all-gather, although only rank 0 consumes the result: on RCCL a gather is a series of
    # point-to-point receives into rank 0, an all-gather one collective that uses every xGMI link
    # (the logits are 2 * vocab / tp bytes per sequence and rank: 9.7 MB per rank at 256 sequences
    # and tp = 8). The other ranks drop their copy.
    parts = [torch.empty_like(t) for _ in range(tp_size())]
    dist.all_gather(parts, t.contiguous())
    return parts if tp_rank() == 0 else None


def all_gather(t):
    """Every rank receives every rank's tensor, stacked in rank order on a new first axis. For the
    few bytes per sequence of the sharded greedy pick (ParallelLMHead.greedy_tokens)."""
    if _CARRIER is not None:
        return _CARRIER.all_gather(t)
    parts = [torch.empty_like(t) for _ in range(tp_size())]
    dist.all_gather(parts, t)
    return torch.stack(parts)


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0, f"{numerator} is not divisible by {denominator}"
    return numerator // denominator
