"""The persistent form of the long-prompt prefill kernel (lv_prefill_attn with an item table,
tile_order_lanes > 0): one workgroup per compute unit walks a host-balanced list of query tiles,
the next tile's Q rows and first K/V tiles arriving while the current one is computed. Per tile it
runs the arithmetic of the one-tile-per-workgroup launch, so the two must agree bit for bit, packed
and paged, on full, ragged, tiny and empty-lane batches; and both against the CPU oracle."""
import pytest
import torch

from oracle.attention_oracle import prefill_ref

pytestmark = pytest.mark.gpu
TOL = 2e-2
D, BLOCK = 128, 256


def _packed(gpu, lens, Hq, Hkv, seed):
    g = torch.Generator().manual_seed(seed)
    total = sum(lens)
    q = torch.randn(total, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(total, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(total, Hkv, D, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32)
    return q, k, v, cu


def _run(gpu, q, k, v, cu_q, cu_k, lens_q, lens_k, persistent, block_table=None):
    from light_vllm_amd import ops
    Hq, Hkv = q.size(1), k.size(-2)
    order = ops.prefill_tile_order(lens_q, lens_k, Hq, Hkv, gpu, kernel="pp", paged=block_table is not None,
                                   persistent=persistent, num_splits=1) if persistent else None
    if persistent:
        assert order is not None and order.lanes > 0
    before = ops.STATS["tile_order_dropped"]
    out = torch.full((q.size(0), Hq, D), float("nan"), dtype=torch.bfloat16, device=gpu)
    ops.prefill_attention(q.to(gpu), k.to(gpu), v.to(gpu), cu_q.to(gpu), cu_k.to(gpu), max(lens_q), max(lens_k), D ** -0.5,
                          block_table=None if block_table is None else block_table.to(gpu), out=out, kernel="pp",
                          num_splits=1, tile_order=order)
    torch.cuda.synchronize()
    assert ops.STATS["tile_order_dropped"] == before, "the item table was not used"
    return out


@pytest.mark.parametrize("lens,Hq,Hkv", [
    ([4096] * 4, 32, 8),                 # 8 lanes x 8 tiles each
    ([4096, 3000, 1024, 8264], 32, 8),   # ragged: lists of unequal depth
    ([4096] * 4, 8, 1),                  # 8 columns x 4 lanes
    ([2500, 2100], 4, 1),                # 8 columns x 8 lanes, most lanes empty
    ([300, 100, 257, 1], 16, 8),         # tiles of one or two key tiles, partial rows
    ([64, 65, 129], 8, 2),
])
def test_persistent_equals_one_tile_per_workgroup(gpu, lens, Hq, Hkv):
    q, k, v, cu = _packed(gpu, lens, Hq, Hkv, seed=len(lens) * 7 + Hq)
    a = _run(gpu, q, k, v, cu, cu, lens, lens, persistent=False)
    b = _run(gpu, q, k, v, cu, cu, lens, lens, persistent=True)
    assert torch.isfinite(b.float()).all()
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    if sum(lens) <= 6000:
        ref = prefill_ref(q, k, v, cu, cu, D ** -0.5)
        assert (b.float().cpu() - ref).abs().max().item() <= TOL


@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1)])
def test_persistent_paged_prefix(gpu, Hq, Hkv):
    """Prefix-cached prefill: new rows over cached + new keys read through the block table."""
    lens_k = [3000, 2304, 700, 2600]
    cached = [1024, 2048, 0, 512]
    lens_q = [a - c for a, c in zip(lens_k, cached)]
    g = torch.Generator().manual_seed(5)
    need = [-(-n // BLOCK) for n in lens_k]
    nblk = sum(need) + 3
    kc = torch.randn(nblk, BLOCK, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(nblk, BLOCK, Hkv, D, generator=g).to(torch.bfloat16)
    perm = torch.randperm(nblk, generator=g).to(torch.int32)
    bt = torch.full((len(lens_k), max(need)), -1, dtype=torch.int32)
    pos = 0
    for i, n in enumerate(need):
        bt[i, :n] = perm[pos:pos + n]
        pos += n
    q = torch.randn(sum(lens_q), Hq, D, generator=g).to(torch.bfloat16)
    cu_q = torch.tensor([0] + torch.tensor(lens_q).cumsum(0).tolist(), dtype=torch.int32)
    cu_k = torch.tensor([0] + torch.tensor(lens_k).cumsum(0).tolist(), dtype=torch.int32)
    a = _run(gpu, q, kc, vc, cu_q, cu_k, lens_q, lens_k, persistent=False, block_table=bt)
    b = _run(gpu, q, kc, vc, cu_q, cu_k, lens_q, lens_k, persistent=True, block_table=bt)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    ref = prefill_ref(q, kc, vc, cu_q, cu_k, D ** -0.5, bt)
    assert (b.float().cpu() - ref).abs().max().item() <= TOL


def test_item_table_checks(gpu):
    """An item table built for the other addressing mode, or for another head count, is dropped by
    the wrapper (the call then runs without it); the C entry point rejects a wrong lane count."""
    from light_vllm_amd import _lib, ops
    lens = [2100, 2500]
    q, k, v, cu = _packed(gpu, lens, 8, 2, seed=3)
    want = _run(gpu, q, k, v, cu, cu, lens, lens, persistent=False)
    wrong_mode = ops.prefill_tile_order(lens, lens, 8, 2, gpu, kernel="pp", paged=True, persistent=True, num_splits=1)
    wrong_heads = ops.prefill_tile_order(lens, lens, 16, 2, gpu, kernel="pp", persistent=True, num_splits=1)
    for order in (wrong_mode, wrong_heads):
        before = ops.STATS["tile_order_dropped"]
        out = ops.prefill_attention(q.to(gpu), k.to(gpu), v.to(gpu), cu.to(gpu), cu.to(gpu), max(lens), max(lens), D ** -0.5,
                                    kernel="pp", num_splits=1, tile_order=order)
        assert ops.STATS["tile_order_dropped"] == before + 1
        assert torch.equal(out, want)
    lib = _lib.load()
    good = ops.prefill_tile_order(lens, lens, 8, 2, gpu, kernel="pp", persistent=True, num_splits=1)
    qd, kd, vd, cud = q.to(gpu), k.to(gpu), v.to(gpu), cu.to(gpu)
    out = torch.empty_like(qd)
    rc = lib.lv_prefill_attn(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), qd.stride(0), kd.stride(0),
                             vd.stride(0), out.stride(0), cud.data_ptr(), cud.data_ptr(), 2, max(lens), max(lens), None, 0, 0,
                             8, 2, D, D ** -0.5, qd.size(0), None, 0, 1, good.table.data_ptr(), good.table.numel(), 256,
                             good.lanes + 1, ops.PREFILL_KERNELS["pp"], torch.cuda.current_stream().cuda_stream)
    assert rc != 0 and b"lanes" in lib.lv_last_error()
