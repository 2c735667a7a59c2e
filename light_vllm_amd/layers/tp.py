"""Tensor-parallel group helpers. One process per GPU; the default process group is the TP group
(backend "nccl" = RCCL over xGMI on MI355X, "gloo" in CPU tests). Without an initialised group
every layer behaves as tp_size 1."""
from __future__ import annotations

import torch.distributed as dist


def tp_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def tp_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def divide(numerator: int, denominator: int) -> int:
    assert numerator % denominator == 0, f"{numerator} is not divisible by {denominator}"
    return numerator // denominator
