"""Per-step metadata bus between the model runner and the attention / LM-head layers.

Same contract as the reference's lightvllm/utils/context.py:15-75: a module-global dataclass that
`set_context` replaces wholesale before each forward pass and `reset_context` clears after it.
Field order and defaults match so positional `set_context(...)` calls behave identically; the two
extra fields, `seq_order`, `tile_order` and `decode_plan`, come last and default to None (launch-order
hints and work lists the runner builds from the lengths it has on the host; results never depend on
them beyond fp32 summation order).
Not thread-safe, by design of the contract (one Python thread per rank process).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class Context:
    is_prefill: bool = False
    cu_seqlens_q: Optional[torch.Tensor] = None   # int32 [B+1], prefill only
    cu_seqlens_k: Optional[torch.Tensor] = None   # int32 [B+1], prefill only
    max_seqlen_q: int = 0
    max_seqlen_k: int = 0
    slot_mapping: Optional[torch.Tensor] = None   # int32 [N]
    context_lens: Optional[torch.Tensor] = None   # int32 [B], decode only
    block_tables: Optional[torch.Tensor] = None   # int32 [B, max_blocks], -1 padded
    seq_order: Optional[torch.Tensor] = None      # int32 [B], decode only: launch order, longest first
    tile_order: Optional[object] = None           # prefill only: prefill_order.TileOrder (query-tile launch order)
    decode_plan: Optional[object] = None          # decode only: decode_plan.DecodePlan (work list cutting long sequences)


_CONTEXT = Context()


def get_context() -> Context:
    return _CONTEXT


def set_context(is_prefill, cu_seqlens_q=None, cu_seqlens_k=None, max_seqlen_q=0, max_seqlen_k=0,
                slot_mapping=None, context_lens=None, block_tables=None, seq_order=None,
                tile_order=None, decode_plan=None) -> None:
    global _CONTEXT
    _CONTEXT = Context(is_prefill, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                       slot_mapping, context_lens, block_tables, seq_order, tile_order, decode_plan)


def reset_context() -> None:
    global _CONTEXT
    _CONTEXT = Context()
