"""Switch between the one-launch HIP kernels for the memory-bound steps around attention
(ops.qk_norm_rope_store, ops.add_rms_norm, ops.silu_and_mul) and the op-by-op torch modules that
restate the reference layers. GPU tensors take the HIP kernels unless LV_FUSED=0 (or
`set_enabled(False)`, which the parity tests use to compare the two on the same inputs); CPU
tensors, which only the plumbing tests produce, always take the torch modules."""
from __future__ import annotations

import os

import torch

_ENABLED = os.environ.get("LV_FUSED", "1") != "0"


def set_enabled(flag: bool) -> bool:
    global _ENABLED
    old, _ENABLED = _ENABLED, bool(flag)
    return old


def use_fused(x: torch.Tensor) -> bool:
    return _ENABLED and x.is_cuda and x.dtype == torch.bfloat16
