"""Caches larger than 2^31 elements (a KV cache sized for 288 GB holds ~1.5 G elements per layer
and tensor): block ids near the top of such a cache must address correctly in every kernel that
takes cache rows — store, fused norm/rope/store, paged decode (both kernels), paged prefill.
The same data placed in low blocks of a small cache is the reference: outputs bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu
D, BS, HKV = 128, 256, 8
NBLK = 8200                      # 8200 * 256 * 8 * 128 = 2.15 G elements > 2^31


@pytest.fixture(scope="module")
def caches(gpu):
    big_k = torch.zeros(NBLK, BS, HKV, D, dtype=torch.bfloat16, device=gpu)
    big_v = torch.zeros(NBLK, BS, HKV, D, dtype=torch.bfloat16, device=gpu)
    assert big_k.numel() > 2 ** 31
    yield big_k, big_v
    del big_k, big_v
    torch.cuda.empty_cache()


def _tables(lens, top):
    """The same logical blocks at the top of the big cache and at the bottom of a small one."""
    need = [(L + BS - 1) // BS for L in lens]
    width = max(need)
    hi = torch.full((len(lens), width), -1, dtype=torch.int32)
    lo = torch.full((len(lens), width), -1, dtype=torch.int32)
    n = 0
    for b, c in enumerate(need):
        for j in range(c):
            hi[b, j] = top - n
            lo[b, j] = n
            n += 1
    return hi, lo, n


@pytest.mark.parametrize("Hq", [16, 32])
def test_decode_reads_top_blocks(gpu, caches, monkeypatch, Hq):
    from light_vllm_amd import ops
    big_k, big_v = caches
    g = torch.Generator().manual_seed(5)
    lens = [300, 513, 70, 1024]
    hi, lo, n = _tables(lens, NBLK - 1)
    small_k = torch.randn(n, BS, HKV, D, generator=g).to(torch.bfloat16).to(gpu)
    small_v = torch.randn(n, BS, HKV, D, generator=g).to(torch.bfloat16).to(gpu)
    for i in range(n):
        big_k[NBLK - 1 - i] = small_k[i]
        big_v[NBLK - 1 - i] = small_v[i]
    q = torch.randn(len(lens), Hq, D, generator=g).to(torch.bfloat16).to(gpu)
    ctx = torch.tensor(lens, dtype=torch.int32, device=gpu)
    for ns in (1, 3):
        ref = ops.decode_attention(q, small_k, small_v, ctx, lo.to(gpu), D ** -0.5, num_splits=ns)
        out = ops.decode_attention(q, big_k, big_v, ctx, hi.to(gpu), D ** -0.5, num_splits=ns)
        assert torch.equal(out, ref)


def test_prefill_reads_top_blocks(gpu, caches):
    from light_vllm_amd import ops
    big_k, big_v = caches
    g = torch.Generator().manual_seed(6)
    lens_k, lens_q = [600, 300], [88, 300]          # the first sequence has 512 cached tokens
    hi, lo, n = _tables(lens_k, NBLK - 1)
    small_k = torch.randn(n, BS, HKV, D, generator=g).to(torch.bfloat16).to(gpu)
    small_v = torch.randn(n, BS, HKV, D, generator=g).to(torch.bfloat16).to(gpu)
    for i in range(n):
        big_k[NBLK - 1 - i] = small_k[i]
        big_v[NBLK - 1 - i] = small_v[i]
    q = torch.randn(sum(lens_q), 32, D, generator=g).to(torch.bfloat16).to(gpu)
    cu_q = torch.tensor([0, 88, 388], dtype=torch.int32, device=gpu)
    cu_k = torch.tensor([0, 600, 900], dtype=torch.int32, device=gpu)
    ref = ops.prefill_attention(q, small_k, small_v, cu_q, cu_k, 300, 600, D ** -0.5, block_table=lo.to(gpu))
    out = ops.prefill_attention(q, big_k, big_v, cu_q, cu_k, 300, 600, D ** -0.5, block_table=hi.to(gpu))
    assert torch.equal(out, ref)


def test_stores_write_top_slots(gpu, caches):
    from light_vllm_amd import ops
    big_k, big_v = caches
    g = torch.Generator().manual_seed(7)
    N = 40
    k = torch.randn(N, HKV, D, generator=g).to(torch.bfloat16).to(gpu)
    v = torch.randn(N, HKV, D, generator=g).to(torch.bfloat16).to(gpu)
    top_slot = NBLK * BS - 1
    slots = torch.tensor([top_slot - 3 * i for i in range(N)], dtype=torch.int32, device=gpu)
    assert int(slots.min()) * HKV * D > 2 ** 31
    big_k[NBLK - 1].zero_()
    big_v[NBLK - 1].zero_()
    ops.store_kvcache(k, v, big_k, big_v, slots)
    flat_k, flat_v = big_k.view(-1, HKV, D), big_v.view(-1, HKV, D)
    assert torch.equal(flat_k[slots.long()], k) and torch.equal(flat_v[slots.long()], v)
    # fused norm + rope + store: the stored K rows must equal the in-place rotated k
    big_k[NBLK - 1].zero_()
    big_v[NBLK - 1].zero_()
    q = torch.randn(N, 32, D, generator=g).to(torch.bfloat16).to(gpu)
    k2 = k.clone()
    pos = torch.arange(N, device=gpu)
    w = torch.ones(D, dtype=torch.bfloat16, device=gpu)
    cs = torch.randn(64, D, generator=g).to(gpu)
    ops.qk_norm_rope_store(q, k2, v, pos, w, w, 1e-6, cs, big_k, big_v, slots)
    assert torch.equal(flat_k[slots.long()], k2) and torch.equal(flat_v[slots.long()], v)
