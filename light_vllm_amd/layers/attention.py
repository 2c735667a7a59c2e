"""Drop-in replacement for the reference's Attention module (lightvllm/layers/attention.py:111-179).

Same constructor `(num_heads, head_dim, scale, num_kv_heads)` (per-rank head counts), the same two
rebinding attributes `k_cache` / `v_cache` that the model runner discovers with hasattr
(engine/model_runner.py:249-254), and the same `forward(q, k, v)` reading its metadata from the
module-global Context. The three device calls go to hand-written gfx950 kernels through the C ABI.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..utils.context import get_context

store_kvcache = ops.store_kvcache


class Attention(nn.Module):

    def __init__(self, num_heads, head_dim, scale, num_kv_heads):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = scale
        self.num_kv_heads = num_kv_heads
        # placeholders; the runner rebinds them to per-layer views of its KV tensor
        self.k_cache = torch.tensor([])
        self.v_cache = torch.tensor([])

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                kv_stored: bool = False) -> torch.Tensor:
        """kv_stored=True: the caller already scattered this step's K/V rows into the cache
        (ops.qk_norm_rope_store does it in the same launch as the q/k normalisation and rotation)."""
        q = q.view(-1, self.num_heads, self.head_dim)
        k = k.view(-1, self.num_kv_heads, self.head_dim)
        v = v.view(-1, self.num_kv_heads, self.head_dim)
        context = get_context()
        k_cache, v_cache = self.k_cache, self.v_cache

        # The store is stream-ordered before the attention launch below, so a prefix-cached
        # prefill and every decode step read the rows written here (reference :149-150).
        if not kv_stored and k_cache.numel() and v_cache.numel():
            store_kvcache(k, v, k_cache, v_cache, context.slot_mapping)

        if context.is_prefill:
            if context.block_tables is not None:  # some prefix is cached: read K/V from the cache
                k, v = k_cache, v_cache
            o = ops.prefill_attention(q, k, v, context.cu_seqlens_q, context.cu_seqlens_k,
                                      context.max_seqlen_q, context.max_seqlen_k, self.scale,
                                      block_table=context.block_tables, tile_order=context.tile_order)
        else:
            o = ops.decode_attention(q, k_cache, v_cache, context.context_lens,
                                     context.block_tables, self.scale, seq_order=context.seq_order,
                                     plan=context.decode_plan)
        return o.view(-1, self.num_heads * self.head_dim)
