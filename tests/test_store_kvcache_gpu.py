"""store_kvcache parity: HIP kernel vs the CPU oracle, bit-exact (reference attention.py:9-108)."""
import pytest
import torch

from oracle.attention_oracle import store_kvcache_ref
from tests._cases import rand_bf16

pytestmark = pytest.mark.gpu


def _run(gpu, N, Hkv, num_slots, seed, strided_v=True, with_skips=False):
    from light_vllm_amd import ops
    D = 128
    gen = torch.Generator().manual_seed(seed)
    Hq = 2 * Hkv
    qkv = rand_bf16(gen, N, (Hq + 2 * Hkv) * D)          # fused activation, as models/qwen3.py:94-95
    key = rand_bf16(gen, N, Hkv, D)                        # K is a fresh contiguous tensor after RoPE
    value = qkv[:, (Hq + Hkv) * D:].view(N, Hkv, D) if strided_v else rand_bf16(gen, N, Hkv, D)
    slots = torch.randperm(num_slots, generator=gen)[:N].to(torch.int32)
    if with_skips and N > 3:
        slots[1] = -1
        slots[N // 2] = -1
    k_cache = rand_bf16(gen, num_slots // 256, 256, Hkv, D)
    v_cache = rand_bf16(gen, num_slots // 256, 256, Hkv, D)
    k_ref, v_ref = k_cache.clone(), v_cache.clone()
    store_kvcache_ref(key, value, k_ref, v_ref, slots)

    qkv_d = qkv.to(gpu)
    value_d = qkv_d[:, (Hq + Hkv) * D:].view(N, Hkv, D) if strided_v else value.to(gpu)
    k_d, v_d = k_cache.to(gpu), v_cache.to(gpu)
    ops.store_kvcache(key.to(gpu), value_d, k_d, v_d, slots.to(gpu))
    torch.cuda.synchronize()
    assert torch.equal(k_d.cpu().view(torch.int16), k_ref.view(torch.int16))
    assert torch.equal(v_d.cpu().view(torch.int16), v_ref.view(torch.int16))


@pytest.mark.parametrize("N,Hkv", [(1, 8), (256, 8), (1000, 8), (257, 1), (4096, 2), (3, 4)])
def test_store_matches_oracle(gpu, N, Hkv):
    _run(gpu, N, Hkv, num_slots=256 * 20, seed=N + Hkv)


def test_store_contiguous_value(gpu):
    _run(gpu, 512, 8, num_slots=256 * 4, seed=7, strided_v=False)


def test_store_skips_negative_slots(gpu):
    _run(gpu, 64, 8, num_slots=256 * 2, seed=9, with_skips=True)


def test_store_empty(gpu):
    from light_vllm_amd import ops
    k = torch.empty(0, 8, 128, dtype=torch.bfloat16, device=gpu)
    kc = torch.zeros(2, 256, 8, 128, dtype=torch.bfloat16, device=gpu)
    ops.store_kvcache(k, k, kc, kc.clone(), torch.empty(0, dtype=torch.int32, device=gpu))
    torch.cuda.synchronize()
    assert kc.abs().sum().item() == 0


def test_store_round_trip_large(gpu):
    """Full-size property: scatter N rows through a random permutation, gather them back."""
    from light_vllm_amd import ops
    N, Hkv, D = 16384, 8, 128                     # the 4x4096 prefill batch of Qwen3-8B
    gen = torch.Generator().manual_seed(3)
    key = rand_bf16(gen, N, Hkv, D).to(gpu)
    value = rand_bf16(gen, N, Hkv, D).to(gpu)
    num_slots = 256 * 80
    slots = torch.randperm(num_slots, generator=gen)[:N].to(torch.int32).to(gpu)
    kc = torch.zeros(80, 256, Hkv, D, dtype=torch.bfloat16, device=gpu)
    vc = torch.zeros_like(kc)
    ops.store_kvcache(key, value, kc, vc, slots)
    torch.cuda.synchronize()
    idx = slots.long()
    assert torch.equal(kc.view(-1, Hkv, D)[idx], key)
    assert torch.equal(vc.view(-1, Hkv, D)[idx], value)
    untouched = torch.ones(num_slots, dtype=torch.bool, device=gpu)
    untouched[idx] = False
    assert kc.view(num_slots, -1)[untouched].abs().sum().item() == 0
