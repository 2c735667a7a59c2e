"""CPU oracle for the attention hot path. TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (light_vllm_amd) never does and has no CPU path of its own.

What it restates
  store_kvcache_ref   the Triton scatter at lightvllm/layers/attention.py:35-52 (+ layout asserts :90-97)
  prefill_ref         the call flash_attn_varlen_func(..., causal=True, block_table) at attention.py:159-165
  decode_ref          the call flash_attn_with_kvcache(..., cache_seqlens, block_table) at attention.py:172-175
  attention_forward_ref  the dispatcher Attention.forward, attention.py:126-179

Parity status. The attention arithmetic itself lives in the third-party `flash_attn` package, which
the reference neither vendors nor pins (pyproject.toml:5-15 lists no dependencies) and which is
not installed here; the reference holds no test or golden vector for this path. The two attention
functions below therefore restate flash-attention's published semantics (softmax(scale*QK^T)V in
fp32, grouped-query head mapping h -> h // (Hq/Hkv), causal mask aligned to the bottom-right corner
when Lk > Lq, cache_seqlens counting the newest token) anchored on the reference's call sites:
PARITY UNPINNED for prefill_ref/decode_ref outputs. store_kvcache_ref is a pure copy and is exact
by construction. Host-side logic (BlockManager, Sequence, prepare_*) is pinned separately by golden
fixtures produced from the reference's own modules (oracle/gen_golden.py -> tests/golden/).

All math is fp32 on CPU tensors; inputs may be bf16 and are upcast exactly.
"""
from __future__ import annotations

from typing import Optional

import torch


def store_kvcache_ref(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                      v_cache: torch.Tensor, slot_mapping: torch.Tensor) -> None:
    """In place: cache.view(-1, Hkv*D)[slot_mapping[i]] = x[i] for i in order.

    Follows attention.py:47-52 (one row of D = Hkv*head_dim elements per token at slot*D).
    Slots < 0 or past the cache are skipped: the reference has no such guard, but never produces
    them in eager mode; the HIP kernel uses -1 as its hipGraph padding sentinel.
    """
    N, H, Dh = key.shape
    D = H * Dh
    kc = k_cache.view(-1, D)
    vc = v_cache.view(-1, D)
    slots = slot_mapping.to(torch.int64)
    ok = (slots >= 0) & (slots < kc.size(0))
    rows = torch.nonzero(ok).flatten()
    # index_copy_ with duplicate indices is order-dependent; the kernel's contract for duplicates
    # is "unspecified winner", so tests never feed duplicates.
    kc.index_copy_(0, slots[rows], key.reshape(N, D)[rows].to(kc.dtype))
    vc.index_copy_(0, slots[rows], value.reshape(N, D)[rows].to(vc.dtype))


def gather_paged(cache: torch.Tensor, block_table_row: torch.Tensor, length: int) -> torch.Tensor:
    """Rows [0, length) of one sequence from a paged cache [nblk, bs, Hkv, D] -> [length, Hkv, D]."""
    bs = cache.size(1)
    nb = (length + bs - 1) // bs
    idx = block_table_row[:nb].to(torch.int64)
    return cache[idx].reshape(nb * bs, cache.size(2), cache.size(3))[:length]


def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, causal_offset: Optional[int]) -> torch.Tensor:
    """q [Lq,Hq,D], k/v [Lk,Hkv,D] fp32 -> [Lq,Hq,D] fp32. causal_offset = Lk - Lq (None = no mask)."""
    Lq, Hq, D = q.shape
    Lk, Hkv, _ = k.shape
    G = Hq // Hkv
    qg = q.reshape(Lq, Hkv, G, D)
    s = torch.einsum("qhgd,khd->hgqk", qg, k) * scale
    if causal_offset is not None:
        r = torch.arange(Lq).unsqueeze(1)
        t = torch.arange(Lk).unsqueeze(0)
        s = s.masked_fill(~(t <= r + causal_offset), float("-inf"))
    # rows that see no key (possible only when Lk < Lq) produce zeros
    m = s.amax(dim=-1, keepdim=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m)
    den = p.sum(dim=-1, keepdim=True)
    p = torch.where(den > 0, p / den.clamp_min(1e-38), torch.zeros_like(p))
    o = torch.einsum("hgqk,khd->qhgd", p, v)
    return o.reshape(Lq, Hq, D)


def prefill_ref(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cu_seqlens_q: torch.Tensor,
                cu_seqlens_k: torch.Tensor, scale: float,
                block_table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Varlen causal attention. q [total_q,Hq,D]; packed k/v [total_k,Hkv,D] or, with block_table,
    paged caches [nblk,bs,Hkv,D]. Returns fp32 [total_q,Hq,D]."""
    cq = cu_seqlens_q.tolist()
    ck = cu_seqlens_k.tolist()
    out = torch.zeros(q.shape, dtype=torch.float32)
    for b in range(len(cq) - 1):
        Lq, Lk = cq[b + 1] - cq[b], ck[b + 1] - ck[b]
        if Lq == 0:
            continue
        qb = q[cq[b]:cq[b + 1]].float()
        if block_table is None:
            kb = k[ck[b]:ck[b + 1]].float()
            vb = v[ck[b]:ck[b + 1]].float()
        else:
            kb = gather_paged(k, block_table[b], Lk).float()
            vb = gather_paged(v, block_table[b], Lk).float()
        if Lk == 0:
            continue
        out[cq[b]:cq[b + 1]] = _attend(qb, kb, vb, scale, Lk - Lq)
    return out


def decode_ref(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
               context_lens: torch.Tensor, block_table: torch.Tensor, scale: float) -> torch.Tensor:
    """q [B,Hq,D]; caches [nblk,bs,Hkv,D]; returns fp32 [B,Hq,D]. Rows with context_len 0 are zero."""
    B = q.size(0)
    out = torch.zeros(q.shape, dtype=torch.float32)
    lens = context_lens.tolist()
    for b in range(B):
        L = lens[b]
        if L <= 0:
            continue
        kb = gather_paged(k_cache, block_table[b], L).float()
        vb = gather_paged(v_cache, block_table[b], L).float()
        out[b:b + 1] = _attend(q[b:b + 1].float(), kb, vb, scale, None)
    return out


def attention_forward_ref(q, k, v, k_cache, v_cache, *, num_heads, num_kv_heads, head_dim, scale,
                          is_prefill, slot_mapping, cu_seqlens_q=None, cu_seqlens_k=None,
                          context_lens=None, block_tables=None) -> torch.Tensor:
    """The Attention.forward dispatcher (attention.py:126-179) on CPU tensors: store, then prefill
    (from the caches when block_tables is given) or decode. Mutates the caches. Returns fp32."""
    q = q.view(-1, num_heads, head_dim)
    k = k.view(-1, num_kv_heads, head_dim)
    v = v.view(-1, num_kv_heads, head_dim)
    if k_cache.numel() and v_cache.numel():
        store_kvcache_ref(k, v, k_cache, v_cache, slot_mapping)
    if is_prefill:
        if block_tables is not None:
            o = prefill_ref(q, k_cache, v_cache, cu_seqlens_q, cu_seqlens_k, scale, block_tables)
        else:
            o = prefill_ref(q, k, v, cu_seqlens_q, cu_seqlens_k, scale)
    else:
        o = decode_ref(q, k_cache, v_cache, context_lens, block_tables, scale)
    return o.view(-1, num_heads * head_dim)
