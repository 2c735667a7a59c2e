"""The CPU oracle against independent statements of the same math: torch SDPA with explicit
masks, hand-computed small cases, and the worked examples in the reference's docstrings."""
import torch
import torch.nn.functional as F

from oracle.attention_oracle import (attention_forward_ref, decode_ref, gather_paged, prefill_ref,
                                     store_kvcache_ref)
from tests._cases import make_decode_case, make_prefill_case


def test_store_known_answer():
    key = torch.arange(2 * 2 * 4, dtype=torch.float32).view(2, 2, 4).to(torch.bfloat16)
    val = -key
    kc = torch.zeros(1, 4, 2, 4, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    store_kvcache_ref(key, val, kc, vc, torch.tensor([3, 1], dtype=torch.int32))
    assert torch.equal(kc[0, 3], key[0]) and torch.equal(kc[0, 1], key[1])
    assert torch.equal(vc[0, 3], val[0]) and kc[0, 0].abs().sum() == 0 and kc[0, 2].abs().sum() == 0


def test_store_skips_out_of_range():
    key = torch.ones(3, 1, 8, dtype=torch.bfloat16)
    kc = torch.zeros(1, 2, 1, 8, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    store_kvcache_ref(key, key, kc, vc, torch.tensor([-1, 1, 2], dtype=torch.int32))
    assert kc[0, 0].abs().sum() == 0 and kc[0, 1].sum() == 8


def test_single_key_returns_value():
    """Softmax over one key is 1: the output is that key's value row (per kv head group)."""
    case = make_decode_case(1, [1], 4, 2, block_size=32)
    out = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
    blk = int(case["block_table"][0, 0])
    v = case["v_cache"][blk, 0].float()           # [Hkv, D]
    assert torch.allclose(out[0, 0], v[0]) and torch.allclose(out[0, 1], v[0])
    assert torch.allclose(out[0, 2], v[1]) and torch.allclose(out[0, 3], v[1])


def test_decode_matches_sdpa():
    case = make_decode_case(2, [70, 5, 33], 8, 2, block_size=32)
    out = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
    for b, L in enumerate(case["context_lens"].tolist()):
        k = gather_paged(case["k_cache"], case["block_table"][b], L).float().permute(1, 0, 2)   # [Hkv, L, D]
        v = gather_paged(case["v_cache"], case["block_table"][b], L).float().permute(1, 0, 2)
        q = case["q"][b].float().view(2, 4, 128)                                               # [Hkv, G, D]
        ref = F.scaled_dot_product_attention(q, k, v, scale=case["scale"])                       # [Hkv, G, D]
        assert torch.allclose(out[b], ref.reshape(8, 128), atol=1e-5, rtol=1e-5)


def test_prefill_matches_sdpa_bottom_right_mask():
    lens_q, lens_k = [5, 40, 1], [37, 40, 9]
    case = make_prefill_case(3, lens_q, lens_k, 4, 2)
    out = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"])
    cq, ck = case["cu_q"].tolist(), case["cu_k"].tolist()
    for b in range(3):
        q = case["q"][cq[b]:cq[b + 1]].float().permute(1, 0, 2)                                # [Hq, Lq, D]
        k = case["k"][ck[b]:ck[b + 1]].float().permute(1, 0, 2).repeat_interleave(2, 0)        # [Hq, Lk, D]
        v = case["v"][ck[b]:ck[b + 1]].float().permute(1, 0, 2).repeat_interleave(2, 0)
        Lq, Lk = lens_q[b], lens_k[b]
        mask = torch.arange(Lk)[None, :] <= torch.arange(Lq)[:, None] + (Lk - Lq)
        ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=case["scale"]).permute(1, 0, 2)
        assert torch.allclose(out[cq[b]:cq[b + 1]], ref, atol=1e-5, rtol=1e-5)


def test_paged_prefill_equals_packed():
    lens = [300, 17]
    paged = make_prefill_case(4, lens, lens, 4, 2, block_size=64, paged=True)
    k = torch.cat([gather_paged(paged["k"], paged["block_table"][b], L) for b, L in enumerate(lens)])
    v = torch.cat([gather_paged(paged["v"], paged["block_table"][b], L) for b, L in enumerate(lens)])
    a = prefill_ref(paged["q"], paged["k"], paged["v"], paged["cu_q"], paged["cu_k"], paged["scale"], paged["block_table"])
    b = prefill_ref(paged["q"], k, v, paged["cu_q"], paged["cu_k"], paged["scale"])
    assert torch.equal(a, b)


def test_last_prefill_row_equals_decode():
    case = make_prefill_case(5, [90], [90], 8, 2, block_size=32, paged=True)
    pre = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"], case["block_table"])
    dec = decode_ref(case["q"][-1:], case["k"], case["v"], torch.tensor([90], dtype=torch.int32), case["block_table"], case["scale"])
    assert torch.allclose(pre[-1:], dec, atol=1e-6)


def test_dispatcher_decode_reads_the_token_it_just_stored():
    """Attention.forward ordering contract (attention.py:149-175): store, then attend."""
    Hq, Hkv, D = 4, 2, 128
    case = make_decode_case(6, [33], Hq, Hkv, block_size=32)
    blk = int(case["block_table"][0, 1])
    slot = torch.tensor([blk * 32 + 0], dtype=torch.int32)       # token index 32 -> second block, offset 0
    k_new = torch.randn(1, Hkv * D).to(torch.bfloat16)
    v_new = torch.randn(1, Hkv * D).to(torch.bfloat16)
    kc, vc = case["k_cache"].clone(), case["v_cache"].clone()
    out = attention_forward_ref(case["q"].view(1, -1), k_new, v_new, kc, vc, num_heads=Hq, num_kv_heads=Hkv,
                                head_dim=D, scale=case["scale"], is_prefill=False, slot_mapping=slot,
                                context_lens=case["context_lens"], block_tables=case["block_table"])
    assert torch.equal(kc[blk, 0].view(-1), k_new.view(-1))
    manual = decode_ref(case["q"], kc, vc, case["context_lens"], case["block_table"], case["scale"])
    assert torch.equal(out.view(1, Hq, D), manual)


# ---------------------------------------------------------------------------------------------
# layer-step oracle (RMSNorm, add+RMSNorm, RoPE) against outputs of the reference's own modules
# ---------------------------------------------------------------------------------------------
def _layers_golden():
    import os
    from safetensors import safe_open
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "layers_small.safetensors")
    with safe_open(path, "pt") as f:
        meta = f.metadata()
        return {k: f.get_tensor(k) for k in f.keys()}, meta


def _bits(t):
    return t.contiguous().view(torch.int16)


def test_layers_oracle_matches_reference_modules_bitwise():
    from oracle import layers_oracle as lo
    g, meta = _layers_golden()
    eps = float(meta["eps"])
    table = lo.cos_sin_table(128, int(meta["max_position"]), float(meta["base"]))
    assert torch.equal(table[g["positions"]], g["cos_sin_rows"])
    assert torch.equal(_bits(lo.rms_norm_ref(g["q"], g["q_weight"], eps)), _bits(g["q_normed"]))
    assert torch.equal(_bits(lo.rms_norm_ref(g["k"], g["k_weight"], eps)), _bits(g["k_normed"]))
    q_rot, k_rot = lo.qk_norm_rope_ref(g["q"], g["k"], g["positions"], g["q_weight"], g["k_weight"], eps, table)
    assert torch.equal(_bits(q_rot), _bits(g["q_rot"]))
    assert torch.equal(_bits(k_rot), _bits(g["k_rot"]))
    assert torch.equal(_bits(lo.rms_norm_ref(g["x"], g["weight"], eps)), _bits(g["rms_out"]))
    out, res = lo.add_rms_norm_ref(g["x"], g["residual"], g["weight"], eps)
    assert torch.equal(_bits(out), _bits(g["add_out"]))
    assert torch.equal(_bits(res), _bits(g["add_res"]))


def test_package_layer_modules_match_the_layers_oracle():
    """The torch modules the CPU plumbing path uses are the same arithmetic as the oracle."""
    from light_vllm_amd.layers.layernorm import RMSNorm
    from light_vllm_amd.layers.rotary_embedding import RotaryEmbedding
    from oracle import layers_oracle as lo
    g, meta = _layers_golden()
    eps = float(meta["eps"])
    norm = RMSNorm(1024, eps)
    norm.weight.data = g["weight"].clone()
    out, res = norm(g["x"].clone(), g["residual"].clone())
    assert torch.equal(_bits(out), _bits(g["add_out"])) and torch.equal(_bits(res), _bits(g["add_res"]))
    rope = RotaryEmbedding(128, 128, int(meta["max_position"]), float(meta["base"]))
    N = g["q"].size(0)
    q_rot, k_rot = rope(g["positions"], g["q_normed"].view(N, -1), g["k_normed"].view(N, -1))
    assert torch.equal(_bits(q_rot.view_as(g["q_rot"])), _bits(g["q_rot"]))
    assert torch.equal(_bits(k_rot.view_as(g["k_rot"])), _bits(g["k_rot"]))
    x = g["x"][:, :512]
    assert torch.equal(_bits(lo.silu_and_mul_ref(x)), _bits(torch.nn.functional.silu(x[:, :256]) * x[:, 256:]))
