"""SwiGLU: silu(gate) * up over the two halves of the last dim (reference "native" backend,
lightvllm/layers/activation.py:7-29)."""
import torch
import torch.nn.functional as F
from torch import nn

from .. import ops
from .fused import use_fused


class SiluAndMul(nn.Module):

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if use_fused(x):
            return ops.silu_and_mul(x)
        gate, up = torch.chunk(x, 2, dim=-1)
        return F.silu(gate) * up
