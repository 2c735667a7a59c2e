"""Neox-style rotary position embedding with a precomputed cos/sin table
(reference: lightvllm/layers/rotary_embedding.py:5-131): rotate the two halves of each head in
fp32, round once to the activation dtype. rope_scaling is not supported, as in the reference (:158)."""
from __future__ import annotations

import torch
from torch import nn


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    cos = cos.unsqueeze(-2)
    sin = sin.unsqueeze(-2)
    x1, x2 = torch.chunk(x.to(torch.float32), 2, -1)
    y1 = x1 * cos - x2 * sin
    y2 = x2 * cos + x1 * sin
    return torch.cat((y1, y2), -1).to(x.dtype)


class RotaryEmbedding(nn.Module):

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float):
        super().__init__()
        assert rotary_dim == head_size
        self.head_size = head_size
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        self.register_buffer("cos_sin_cache", torch.cat((freqs.cos(), freqs.sin()), -1), persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        n = positions.size(0)
        cos, sin = self.cos_sin_cache[positions].chunk(2, -1)
        q_shape, k_shape = query.shape, key.shape
        query = apply_rotary_emb(query.view(n, -1, self.head_size), cos, sin).view(q_shape)
        key = apply_rotary_emb(key.view(n, -1, self.head_size), cos, sin).view(k_shape)
        return query, key


_ROPE_CACHE: dict = {}


def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float, rope_scaling=None):
    assert rope_scaling is None, "rope_scaling is not supported"
    key = (head_size, rotary_dim, max_position, float(base), str(torch.get_default_device()))
    if key not in _ROPE_CACHE:
        _ROPE_CACHE.clear()   # one live table, like the reference's lru_cache(maxsize=1)
        _ROPE_CACHE[key] = RotaryEmbedding(head_size, rotary_dim, max_position, base)
    return _ROPE_CACHE[key]
