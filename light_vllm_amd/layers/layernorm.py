"""RMSNorm and fused residual-add + RMSNorm with the reference's rounding points
(lightvllm/layers/layernorm.py:28-80): fp32 math, one rounding to the activation dtype before the
weight multiply, the new residual rounded once from the fp32 sum."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .fused import use_fused


class RMSNorm(nn.Module):

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.hidden_size = hidden_size
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def rms_forward(self, x: torch.Tensor) -> torch.Tensor:
        dtype = x.dtype
        xf = x.to(torch.float32)
        var = xf.pow(2).mean(-1, True)
        xf = xf * torch.rsqrt(var + self.eps)
        return xf.to(dtype).mul_(self.weight)

    def add_rms_forward(self, x: torch.Tensor, residual: torch.Tensor):
        dtype = x.dtype
        xf = x.to(torch.float32) + residual.to(torch.float32)
        residual = xf.to(dtype)
        var = xf.pow(2).mean(-1, True)
        xf = xf * torch.rsqrt(var + self.eps)
        return xf.to(dtype).mul_(self.weight), residual

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if use_fused(x):    # one HIP launch; the residual is updated in place
            out = ops.add_rms_norm(x, residual, self.weight, self.eps)
            return out if residual is None else (out, residual)
        if residual is None:
            return self.rms_forward(x)
        return self.add_rms_forward(x, residual)
