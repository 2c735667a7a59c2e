"""TEST-ONLY substitution of the three HIP entry points by the CPU oracle, so the host stack
(Attention dispatcher, ModelRunner, Scheduler, TP layers) can be exercised without a GPU
(BASELINE config 1: "plumbing, no GPU"). The product package has no such path."""
from __future__ import annotations

from contextlib import contextmanager

import torch

from oracle import attention_oracle as orc


def _store(key, value, k_cache, v_cache, slot_mapping):
    orc.store_kvcache_ref(key, value, k_cache, v_cache, slot_mapping)


def _prefill(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, softmax_scale,
             block_table=None, out=None, tile_order=None):
    return orc.prefill_ref(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, block_table).to(q.dtype)


def _decode(q, k_cache, v_cache, context_lens, block_table, softmax_scale, num_splits=0, out=None,
            seq_order=None, plan=None):
    return orc.decode_ref(q, k_cache, v_cache, context_lens, block_table, softmax_scale).to(q.dtype)


@contextmanager
def cpu_kernels():
    from light_vllm_amd import ops
    from light_vllm_amd.layers import attention as attn_mod
    saved = (ops.store_kvcache, ops.prefill_attention, ops.decode_attention, attn_mod.store_kvcache)
    ops.store_kvcache, ops.prefill_attention, ops.decode_attention = _store, _prefill, _decode
    attn_mod.store_kvcache = _store
    try:
        yield
    finally:
        ops.store_kvcache, ops.prefill_attention, ops.decode_attention, attn_mod.store_kvcache = saved
