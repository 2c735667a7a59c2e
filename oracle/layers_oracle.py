"""CPU restatement of the memory-bound layer steps either side of the attention call, with the
reference's rounding points. TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke()
and oracle/ scripts; nothing in light_vllm_amd imports it.

Pinned by tests/golden/layers_small.safetensors, which oracle/gen_layers_golden.py produced by
running the reference's own RMSNorm and RotaryEmbedding modules (eager, CPU) on seeded inputs;
tests/test_oracle_cpu.py checks this file against those outputs bit for bit. silu_and_mul has no
importable reference module here (lightvllm/layers/activation.py imports its CUDA/Triton kernels
at module import) and is restated from lightvllm/layers/activation.py:7-29.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def rms_norm_ref(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """lightvllm/layers/layernorm.py:28-50: fp32 normalisation, round to x.dtype, times weight."""
    dtype = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, True)
    xf = xf * torch.rsqrt(var + eps)
    return xf.to(dtype) * weight


def add_rms_norm_ref(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """lightvllm/layers/layernorm.py:52-80: fp32 sum, new residual = the sum rounded once, the
    variance taken from the unrounded sum."""
    dtype = x.dtype
    xf = x.to(torch.float32) + residual.to(torch.float32)
    new_residual = xf.to(dtype)
    var = xf.pow(2).mean(-1, True)
    xf = xf * torch.rsqrt(var + eps)
    return xf.to(dtype) * weight, new_residual


def cos_sin_table(head_dim: int, max_position: int, base: float) -> torch.Tensor:
    """lightvllm/layers/rotary_embedding.py:76-91: fp32 [max_position, head_dim], cos | sin."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    t = torch.arange(max_position, dtype=torch.float)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1)


def rope_ref(positions: torch.Tensor, x: torch.Tensor, cos_sin: torch.Tensor) -> torch.Tensor:
    """lightvllm/layers/rotary_embedding.py:5-42, 93-131: x [N, H, D]; neox pairing (i, i + D/2);
    fp32 products and sums as separate roundings, one rounding back to x.dtype."""
    cos, sin = cos_sin[positions].chunk(2, dim=-1)
    cos, sin = cos.unsqueeze(-2), sin.unsqueeze(-2)
    x1, x2 = torch.chunk(x.to(torch.float32), 2, dim=-1)
    y1 = x1 * cos - x2 * sin
    y2 = x2 * cos + x1 * sin
    return torch.cat((y1, y2), dim=-1).to(x.dtype)


def qk_norm_rope_ref(q, k, positions, q_weight, k_weight, eps, cos_sin):
    """lightvllm/models/qwen3.py:100-109: q_norm / k_norm per head, then the rotary embedding.
    q [N, Hq, D], k [N, Hkv, D] -> (q', k')."""
    qn = rms_norm_ref(q, q_weight, eps)
    kn = rms_norm_ref(k, k_weight, eps)
    return rope_ref(positions, qn, cos_sin), rope_ref(positions, kn, cos_sin)


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """lightvllm/layers/activation.py:7-29."""
    gate, up = torch.chunk(x, 2, dim=-1)
    return F.silu(gate) * up
