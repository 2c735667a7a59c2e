"""Seeded synthetic inputs shared by the CPU and GPU tests (all tensors are created on CPU)."""
from __future__ import annotations

import torch

D = 128


def rand_bf16(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen, dtype=torch.float32) * scale).to(torch.bfloat16)


def make_block_tables(gen, lens, block_size, num_blocks, pad_cols=0):
    """Random non-contiguous physical blocks per sequence; unused columns are -1."""
    need = [(L + block_size - 1) // block_size for L in lens]
    assert sum(need) <= num_blocks
    perm = torch.randperm(num_blocks, generator=gen).tolist()
    width = max(max(need), 1) + pad_cols
    bt = torch.full((len(lens), width), -1, dtype=torch.int32)
    pos = 0
    for b, n in enumerate(need):
        for j in range(n):
            bt[b, j] = perm[pos]
            pos += 1
    return bt


def make_decode_case(seed, lens, Hq, Hkv, block_size=256, extra_blocks=3, pad_cols=0):
    gen = torch.Generator().manual_seed(seed)
    B = len(lens)
    need = sum((L + block_size - 1) // block_size for L in lens)
    nblk = need + extra_blocks
    k_cache = rand_bf16(gen, nblk, block_size, Hkv, D)
    v_cache = rand_bf16(gen, nblk, block_size, Hkv, D)
    q = rand_bf16(gen, B, Hq, D)
    bt = make_block_tables(gen, lens, block_size, nblk, pad_cols)
    ctx = torch.tensor(lens, dtype=torch.int32)
    return dict(q=q, k_cache=k_cache, v_cache=v_cache, context_lens=ctx, block_table=bt,
                scale=D ** -0.5)


def make_prefill_case(seed, lens_q, lens_k, Hq, Hkv, block_size=256, paged=False, extra_blocks=2):
    """lens_k[b] >= lens_q[b]; paged=True puts K/V in a paged cache (prefix-cached prefill)."""
    gen = torch.Generator().manual_seed(seed)
    tq, tk = sum(lens_q), sum(lens_k)
    q = rand_bf16(gen, tq, Hq, D)
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0).tolist()), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0).tolist()), dtype=torch.int32)
    out = dict(q=q, cu_q=cu_q, cu_k=cu_k, max_q=max(lens_q), max_k=max(lens_k), scale=D ** -0.5)
    if not paged:
        out["k"] = rand_bf16(gen, tk, Hkv, D)
        out["v"] = rand_bf16(gen, tk, Hkv, D)
        out["block_table"] = None
    else:
        need = sum((L + block_size - 1) // block_size for L in lens_k)
        nblk = need + extra_blocks
        out["k"] = rand_bf16(gen, nblk, block_size, Hkv, D)
        out["v"] = rand_bf16(gen, nblk, block_size, Hkv, D)
        out["block_table"] = make_block_tables(gen, lens_k, block_size, nblk)
    return out


def to_dev(case, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in case.items()}
